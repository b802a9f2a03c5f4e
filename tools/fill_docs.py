#!/usr/bin/env python3
"""Replace the @R3_*@ placeholders of DESIGN.md / README.md / profiles/README.md with the numbers of
profiles/r03_bench_default.json and the r03 rocprofv3 summaries (run after tools/summarise_profiles.py r03 r03)."""
import csv
import json
import os
import re


def main():
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r03_bench_default.json')))
    rows = list(csv.DictReader(open(os.path.join(ROOT, 'profiles', 'r03_rocprofv3_kernel_stats_serial.csv'))))
    dom = next(r for r in rows if re.search(r'conv_kernel<128, 2, 4, false>', r['Name']))
    serial = json.load(open(os.path.join(ROOT, 'profiles', 'r03_rocprofv3_bench_line_serial.json')))
    tl = d['detect']['test_loop']
    vals = {
        'R3_IPS': '%.1f' % d['value'], 'R3_MS': '%.1f' % d['ms_per_step'], 'R3_MED': '%.1f' % d['median_ms_per_step'],
        'R3_TF': '%.1f' % d['step_tflops_per_gpu'], 'R3_FRAC': '%.1f' % (100 * d['step_frac_of_fp32_mfma_peak']),
        'R3_DOM': '%.1f' % d['roofline']['achieved'], 'R3_DOMFRAC': '%.1f' % (100 * d['roofline']['frac']),
        'R3_ROCPROF': '%.1f' % (float(dom['AverageNs']) / 1e3), 'R3_AVG': '%.1f' % (1e3 * serial['roofline']['avg_launch_ms']),
        'R3_TRAFFIC': '%.1f' % ((d['roofline']['traffic'] or 0) / 1e6),
        'R3_T16': '%.0f' % tl['eval_batch_16'], 'R3_T1': '%.0f' % tl['eval_batch_1'],
        'R3_B1': '%.2f' % d['detect']['batch1_device'], 'R3_B40': '%.2f' % d['detect']['batch40_device'],
        'R3_CPU': '%.2f' % d['cpu_baseline']['value'], 'R3_CPUGF': '%.0f' % d['cpu_baseline']['gflops'],
        'R3_LOADER': '%.0f' % d['loader_inclusive']['value'], 'R3_LOADONLY': '%.0f' % d['loader_inclusive']['loader_only_images_per_sec'],
        'R3_LOADPIL': '%.0f' % d['loader_inclusive']['loader_only_pillow_images_per_sec'],
        'R3_3S': '%.0f' % d['three_scale_train']['value'],
        'R3_MGPU': '%.1f' % d['multi_gpu']['ms_per_step'], 'R3_MGPLAIN': '%.1f' % d['multi_gpu']['plain_ms_per_step_same_loop'],
    }
    for name in ('DESIGN.md', 'README.md', os.path.join('profiles', 'README.md')):
        p = os.path.join(ROOT, name)
        s = open(p).read()
        left = set(re.findall(r'@(R3_[A-Z0-9]+)@', s))
        for k, v in vals.items():
            s = s.replace('@%s@' % k, v)
        missing = set(re.findall(r'@(R3_[A-Z0-9]+)@', s))
        open(p, 'w').write(s)
        print(name, 'filled', sorted(left - missing), 'unknown', sorted(missing))


if __name__ == '__main__':
    main()
