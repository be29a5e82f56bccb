#!/bin/bash
# GPU box: kernel trace + the three PMC passes of the default bench, into gpurun_out/prof_<tag>/
tag=${1:-cur}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out/pmc
python3 bench.py > $out/bench_line.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/trace -o r -- python3 bench.py --no-cpu-baseline > $out/trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$out/pmc/$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $d -o b -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $d.log 2>&1
done
tail -1 $out/bench_line.log | cut -c1-300
ls $out $out/trace $out/pmc
