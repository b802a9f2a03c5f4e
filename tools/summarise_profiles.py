#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh (under gpurun_out/) into the summaries kept in
profiles/:  python tools/summarise_profiles.py r01 r02

  <out>_rocprofv3_kernel_stats_{serial,overlap}.csv   rocprofv3's own --stats table
  <out>_rocprofv3_bench_line_{serial,overlap}.json    the bench line printed by that very run
  <out>_pmc_traffic.json                              FETCH_SIZE / WRITE_SIZE per kernel and launch

FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a 128-byte request as 64
bytes and is doubled here (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r'^void ', '', name)
    name = name.replace('(anonymous namespace)::', '')
    return name.split('(')[0]


def bench_line(path):
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith('{"metric"'):
            return json.loads(ln)
    raise SystemExit('no bench line in ' + path)


def main():
    tag, out = sys.argv[1], sys.argv[2]
    g = os.path.join(ROOT, 'gpurun_out')
    p = os.path.join(ROOT, 'profiles')
    for mode in ('serial', 'overlap'):
        stats = glob.glob(os.path.join(g, '%s_%s' % (tag, mode), '**', '*kernel_stats.csv'), recursive=True)[0]
        shutil.copy(stats, os.path.join(p, '%s_rocprofv3_kernel_stats_%s.csv' % (out, mode)))
        line = bench_line(os.path.join(g, '%s_%s_bench.json' % (tag, mode)))
        json.dump(line, open(os.path.join(p, '%s_rocprofv3_bench_line_%s.json' % (out, mode)), 'w'), indent=1)
        for r in csv.DictReader(open(stats)):
            if re.search(r'conv_kernel<128, 2, [24], false|conv1x1_persist_kernel<128', r['Name']):
                print(mode, 'rocprofv3 dominant kernel: calls', r['Calls'], 'avg us', float(r['AverageNs']) / 1e3,
                      '| bench avg_launch_ms', (line.get('roofline') or {}).get('avg_launch_ms'),
                      '| overlapped', (line.get('roofline_overlapped') or {}).get('avg_launch_ms'))
    traffic = {}
    for which, key, mult in (('fetch', 'fetch_MB_per_launch', 2.0), ('write', 'write_MB_per_launch', 1.0)):
        f = glob.glob(os.path.join(g, '%s_pmc_%s' % (tag, which), '**', '*counter_collection.csv'), recursive=True)[0]
        acc = {}
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
        for k, (n, v) in acc.items():
            d = traffic.setdefault(k, {})
            d['launches_in_2_steps'] = n
            d[key] = round(v * mult / 1024.0 / n, 2)
    fp = os.path.join(g, '%s_fingerprint.txt' % tag)
    if os.path.exists(fp):
        traffic['_source_fingerprint'] = open(fp).read().strip()   # bench.py reports traffic only while the kernels are these
    mf = glob.glob(os.path.join(g, '%s_pmc_mfma' % tag, '**', '*counter_collection.csv'), recursive=True)
    if mf:
        # MfmaUtil per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -- the counter sums over
        # all SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (MI355X_MICROARCH.md, rocprofv3 PMC section)
        acc = {}
        for r in csv.DictReader(open(mf[0])):
            a = acc.setdefault(short(r['Kernel_Name']), {'n': 0})
            a[r['Counter_Name']] = a.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
            a['n'] += 1
        util = {}
        for k, a in acc.items():
            if a.get('GRBM_GUI_ACTIVE'):
                util[k] = dict(mfma_busy_cycles=a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), gui_active=a['GRBM_GUI_ACTIVE'],
                               mfma_util=round(a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (a['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 4))
        json.dump(util, open(os.path.join(p, '%s_pmc_mfma_util.json' % out), 'w'), indent=1)
        for k in util:
            if re.match(r'conv_kernel<128, 2, [24], false|conv1x1_persist_kernel<128', k) or k.startswith('wgrad_kernel<128, 128, true, false'):
                print('MfmaUtil', k, util[k]['mfma_util'])
    traffic['_note'] = ('rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over: bench.py --steps 1 '
                        '--warmup 1 --no-cpu-baseline --profile-steps 0 --no-overlap --no-detect; MB per launch; FETCH_SIZE '
                        'doubled (gfx950 correction); fabric-side requests, Infinity-Cache hits included')
    json.dump(traffic, open(os.path.join(p, '%s_pmc_traffic.json' % out), 'w'), indent=1)
    for k in traffic:
        if re.match(r'conv_kernel<128, 2, [24], false|conv1x1_persist_kernel<128', k) or k.startswith('wgrad_kernel<128, 128, true, false'):
            print(k, traffic[k])


if __name__ == '__main__':
    main()
