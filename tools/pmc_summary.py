#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> profiles/rNN_pmc_summary.json (per kernel: HBM bytes per launch, MFMA utilisation).

usage: pmc_summary.py <dir with one sub-directory per pass> <out.json> [label]
Every */*counter_collection.csv below <dir> is read; counters may be spread over passes (TCC has 4 slots: FETCH_SIZE takes 3,
WRITE_SIZE 2, so they cannot share a pass - MI355X_MICROARCH.md, rocprofv3 PMC slots).  Per kernel (template arguments and
signatures stripped) the mean per dispatch is taken.  Corrections, exactly as that guide's HBM section prescribes:
  * FETCH_SIZE / WRITE_SIZE are in KB;
  * on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (16 B per lane) - the GEMM kernels here read 16 B per
    lane throughout, so bytes_read = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16-byte stores and is taken as is
    (4-byte epilogue stores are uncalibrated: an upper-bound caveat, ratios between builds are unaffected).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * average kernel duration from the kernel-trace stats * 2.4 GHz); one
v_mfma_f32_16x16x4_f32 holds its SIMD's matrix pipe for 32 cycles, so the counter also cross-checks the algorithmic MAC count
(busy cycles / 32 * 1024 MACs).  usage adds a 4th argument: the *kernel_stats.csv of the --kernel-trace --stats run.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict


def source_sha16(root):
    """Hash of the kernel sources the counters were measured on: bench.py recomputes it and marks the counters stale when the sources changed."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, 'fql_amd', 'csrc', '*'))):
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def base(name):
    n = name.split('(')[0].replace('void ', '').strip()
    return re.sub(r'<.*>', '', n)


def main():
    root, out = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else root
    dur = {}   # kernel -> average duration (ns) from a rocprofv3 --kernel-trace --stats csv of the same command
    if len(sys.argv) > 4:
        for r in csv.DictReader(open(sys.argv[4])):
            dur[base(r['Name'])] = float(r['AverageNs'])
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = base(r['Kernel_Name'])
            if not k.startswith('fql_'):
                continue
            a = acc[k][r['Counter_Name']]
            a[0] += float(r['Counter_Value']); a[1] += 1
    kernels = {}
    for k, cs in sorted(acc.items()):
        m = {c: v[0] / v[1] for c, v in cs.items()}
        d = {'dispatches_seen': max(v[1] for v in cs.values())}
        if 'FETCH_SIZE' in m:
            d['fetch_size_kb_raw'] = round(m['FETCH_SIZE'], 1)
        if 'WRITE_SIZE' in m:
            d['write_size_kb'] = round(m['WRITE_SIZE'], 1)
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            d['hbm_bytes_per_launch'] = int((2.0 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            d['mfma_busy_cycles'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'])
            if k in dur:   # MFMA-pipe busy cycles over the SIMD-cycles of the launch at the 2.4 GHz peak clock (1024 SIMDs)
                d['avg_duration_us'] = round(dur[k] / 1e3, 3)
                d['mfma_util'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * dur[k] * 2.4), 4)
            if m.get('GRBM_GUI_ACTIVE', 0) > 0:   # reads high on dispatches this short (MI355X_MICROARCH.md, DVFS give-back): recorded, not used
                d['grbm_gui_active_per_xcd'] = round(m['GRBM_GUI_ACTIVE'] / 8.0)
        for c in ('SQ_BUSY_CYCLES', 'SQ_WAVES', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_INSTS_MFMA', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
            if c in m:
                d[c.lower()] = round(m[c])
        kernels[k] = d
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    json.dump({'source': label, 'source_sha16': source_sha16(repo), 'kernels': kernels}, open(out, 'w'), indent=1)
    for k, d in kernels.items():
        print(k, d)


if __name__ == '__main__':
    main()
