# same-box A/B of two builds of the library: $1 = path of the alternative .so (the default build is B)
# alternates A,B,A,B so that clock drift shows up as spread, not as a difference
for i in 1 2; do
  for tag in A B; do
    if [ $tag = A ]; then export SEUNET_LIB=$PWD/$1; else unset SEUNET_LIB; fi
    timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$tag$i.log 2>&1
    python - <<PY
import json
l=[x for x in open('gpurun_out/ab_$tag$i.log') if x.startswith('{')]
print("$tag$i", "%.2f ms/step" % json.loads(l[-1])['ms_per_step'] if l else open('gpurun_out/ab_$tag$i.log').read()[-800:])
PY
  done
done
