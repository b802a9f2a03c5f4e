"""Summarise a rocprofv3 kernel trace of tools/test_loop_trace.py: the last pass's device timeline (busy fraction, idle gaps and
what follows them, time per kernel).  Usage: test_loop_trace_summary.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:70], r.get('Queue_Id', r.get('Stream_Id', '?'))))
    rows.sort()
    # the passes: split at the biggest idle stretches... simpler: keep the last third of the head-conv launches
    heads = [i for i, r in enumerate(rows) if 'conv0_direct' in r[2]]
    n = len(heads)
    lo = rows[heads[2 * n // 3]][0]
    last = [r for r in rows if r[0] >= lo]
    t0, t1 = last[0][0], max(r[1] for r in last)
    # union of busy intervals
    busy = 0; cur_s, cur_e = last[0][0], last[0][1]
    gaps = []
    for s, e, k, q in last[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, k))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print('last third of the trace: %.1f ms wall, device busy %.1f ms = %.1f %%; %d forwards' % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), len([r for r in last if 'conv0_direct' in r[2]])))
    per = {}
    for s, e, k, q in last:
        a = per.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
    for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
        print('  %-70s %8.2f ms %6d launches' % (k, t / 1e6, c))
    big = {}
    for g, k in gaps:
        if g > 20000:
            a = big.setdefault(k, [0, 0]); a[0] += g; a[1] += 1
    print('idle gaps over 20 us, by the kernel that ends them:')
    for k, (t, c) in sorted(big.items(), key=lambda kv: -kv[1][0])[:10]:
        print('  %-70s %8.2f ms in %d gaps' % (k, t / 1e6, c))
    print('sum of all gaps %.2f ms (%d gaps)' % (sum(g for g, _ in gaps) / 1e6, len(gaps)))


if __name__ == '__main__':
    main()
