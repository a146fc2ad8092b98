# Runs on the GPU box: bench.py (timed region only) at several pass sizes. usage: tools/spp_per_pass.sh spp:steps[:pipelines] ...
set -o pipefail
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  set -- ${cfg//:/ }; A=$1; B=$2; C=${3:-3}
  F=$R/gpurun_out/spp_${A}_$$.json
  RT_BENCH_PIPELINES=$C RT_BENCH_SPP_PER_STEP=$A timeout -k 10 200 python3 $R/bench.py --steps $B --warmup 2 --no-cpu-baseline --no-secondary --no-strong --no-c1 > $F 2> $F.err || { echo FAILED $cfg; tail -3 $F.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('$F').read().strip().splitlines()[-1])
print('spp per pass $A steps $B pipelines $C: %.0f Mrays/s  %.2f ms/step  %.3f G samples/s' % (d['value'], d['ms_per_step'], d['samples_per_sec']/1e9))"
done
