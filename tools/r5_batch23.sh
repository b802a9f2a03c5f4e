#!/bin/bash
# non-temporal output stores of the tile kernels' epilogue: step time A/B, then FETCH_SIZE per launch (one PMC pass each)
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
rm -f $out/r5_b23_ab.txt
for v in default ntout default ntout; do
  if [ $v = default ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$root/tools/_variants/libfv_$v.so; fi
  timeout -k 10 200 python3 $root/bench.py $Q 2> $out/r5_b23.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_b23_ab.txt || exit 1
done
pmc="--steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-overlap --no-detect --no-loader --no-three-scale --no-rccl-rehearsal"
for v in default ntout; do
  if [ $v = default ]; then unset FV_LIB_PATH; else export FV_LIB_PATH=$root/tools/_variants/libfv_$v.so; fi
  rm -rf $out/b23_fetch_$v
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/b23_fetch_$v -o run -- python3 $root/bench.py $pmc > $out/r5_b23_fetch_$v.json 2> $out/r5_b23_fetch_$v.err || exit 1
  python3 - $out/b23_fetch_$v $v <<'P' | tee -a $out/r5_b23_ab.txt
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    k = re.sub(r'^void ', '', r['Kernel_Name']).replace('(anonymous namespace)::', '').split('(')[0]
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
tot = 0.0
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print('%s %-48s n=%4d fetch %8.1f MB/launch' % (sys.argv[2], k[:48], n, v * 2 / 1024 / n))
print(sys.argv[2], 'total fetch per step %.1f GB' % (sum(v for n, v in acc.values()) * 2 / 1024 / 1024 / 2))
P
  rm -rf $out/b23_fetch_$v
done
