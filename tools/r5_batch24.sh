#!/bin/bash
# chunk-major K order of the multi-tap tile launches (option conv_chunk_major): tests, step time A/B, FETCH_SIZE per launch
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd $root
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_slots_gpu.py tests/test_net_gpu.py -x -q -k "not overfits and not bucketed" > $out/r5_b24_tests.log 2>&1 || { tail -40 $out/r5_b24_tests.log; exit 1; }
tail -3 $out/r5_b24_tests.log
cd /tmp && export TMPDIR=/tmp
Q="--steps 20 --warmup 10 --no-cpu-baseline --no-detect --no-loader --no-three-scale --no-rccl-rehearsal --profile-steps 0"
rm -f $out/r5_b24_ab.txt
for v in on off on off; do
  if [ $v = on ]; then unset FV_OPTIONS; else export FV_OPTIONS=conv_chunk_major=0; fi
  timeout -k 10 200 python3 $root/bench.py $Q 2> $out/r5_b24.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['median_ms_per_step'])" | tee -a $out/r5_b24_ab.txt || exit 1
done
pmc="--steps 1 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-overlap --no-detect --no-loader --no-three-scale --no-rccl-rehearsal"
for v in on off; do
  if [ $v = on ]; then unset FV_OPTIONS; else export FV_OPTIONS=conv_chunk_major=0; fi
  rm -rf $out/b24_fetch_$v
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/b24_fetch_$v -o run -- python3 $root/bench.py $pmc > $out/r5_b24_fetch_$v.json 2> $out/r5_b24_fetch_$v.err || exit 1
  python3 - $out/b24_fetch_$v $v <<'P' | tee -a $out/r5_b24_ab.txt
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = {}; shp = {}
for r in csv.DictReader(open(f)):
    k = re.sub(r'^void ', '', r['Kernel_Name']).replace('(anonymous namespace)::', '').split('(')[0]
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
    if k.startswith('conv_kernel<128, 2, 4, false, 128'):
        b = shp.setdefault(int(r['Grid_Size']) // 512, [0, 0.0]); b[0] += 1; b[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
    print('%s %-48s n=%4d fetch %8.1f MB/launch' % (sys.argv[2], k[:48], n, v * 2 / 1024 / n))
for g, (n, v) in sorted(shp.items(), key=lambda kv: -kv[1][1]):
    print('%s   conv_kernel<128> grid %5d workgroups n=%3d fetch %8.1f MB/launch' % (sys.argv[2], g, n, v * 2 / 1024 / n))
print(sys.argv[2], 'total fetch per step %.1f GB' % (sum(v for n, v in acc.values()) * 2 / 1024 / 1024 / 2))
P
  rm -rf $out/b24_fetch_$v
done
