#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, CSV output) into a
small JSON under profiles/.  Units and the gfx950 correction follow
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; FETCH_SIZE reads exactly
half of the bytes of a coalesced streaming read on gfx950, so it is doubled; WRITE_SIZE is exact.

    python tools/summarize_pmc.py gpurun_out/pmc profiles/pmc_simple_tag_1M_f32.json <kernel substring>
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    src, dst, needle = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch = per_kernel(src + '/fetch_counter_collection.csv', 'FETCH_SIZE')
    write = per_kernel(src + '/write_counter_collection.csv', 'WRITE_SIZE')
    out = {'units': 'bytes per launch', 'fetch_correction': 'FETCH_SIZE KiB * 1024 * 2 (gfx950)',
           'write_correction': 'WRITE_SIZE KiB * 1024', 'kernels': {}}
    for k in fetch:
        if needle in k and k in write:
            f, n = fetch[k]
            w, _ = write[k]
            out['kernels'][k] = {'launches': n, 'fetch_bytes': f * 1024 * 2, 'write_bytes': w * 1024,
                                 'traffic_bytes': f * 1024 * 2 + w * 1024}
    # optional third pass: SQ issue counters (quad-cycle units, summed over the chip) + GRBM_GUI_ACTIVE
    # (summed over the 8 XCDs): VALU utilisation = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * kernel cycles)
    import os
    sq_path = src + '/sq_counter_collection.csv'
    if os.path.exists(sq_path):
        names = ['SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY',
                 'SQ_ACTIVE_INST_ANY', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE']
        per = {n: per_kernel(sq_path, n) for n in names}
        for k in out['kernels']:
            if k not in per['GRBM_GUI_ACTIVE']:
                continue
            c = {n: per[n][k][0] for n in names if k in per[n]}
            cycles = c['GRBM_GUI_ACTIVE'] / 8.0
            c['kernel_cycles'] = cycles
            c['valu_util'] = c['SQ_ACTIVE_INST_VALU'] * 4.0 / (1024.0 * cycles)
            c['wave_wait_frac'] = c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']
            out['kernels'][k]['sq'] = c
    json.dump(out, open(dst, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
