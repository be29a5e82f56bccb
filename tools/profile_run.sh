#!/bin/bash
# GPU box: bench lines + kernel traces + the three PMC passes of the default (fp32 training) bench, into gpurun_out/prof_<tag>/
# usage (through gpurun): bash tools/profile_run.sh <tag> <commit>
tag=${1:-cur}
head=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the per-kernel traces and counters are taken on ONE stream (RCF_SINGLE_STREAM=1 on the --graph 0 commands below): a kernel's
# duration and counters are its own, as in bench.py's roofline events; the bench lines themselves run the default (side stream on)
out=gpurun_out/prof_$tag
mkdir -p $out/pmc
python3 bench.py --steps 20 --warmup 5 > $out/bench_line.log 2> $out/bench_line.err
python3 bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_bf16.log 2>&1
python3 bench.py --dtype f32_3plane --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_f32_3plane.log 2>&1
python3 bench.py --workload infer --steps 20 --warmup 5 > $out/bench_infer.log 2>&1
python3 bench.py --workload radarnet --steps 10 --warmup 3 > $out/bench_radarnet.log 2>&1
python3 bench.py --workload infer --dtype f32 --steps 10 --warmup 3 > $out/bench_infer_f32.log 2>&1
python3 bench.py --workload radarnet --dtype f32 --steps 10 --warmup 3 > $out/bench_radarnet_f32.log 2>&1
# the traced / counted fp32 runs are the HEADLINE step alone (--no-side-leg --no-other-configs: the default line's extra legs would mix four
# workloads into one kernel table)
# eager launches under the profiler (a replayed hipGraph hides the per-launch events bench.py's roofline uses)
RCF_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/trace_$tag -o r -- python3 bench.py --graph 0 --steps 5 --warmup 3 --preheat-s 0 --no-cpu-baseline --no-side-leg --no-other-configs > $out/trace.log 2>&1
grep "^{" $out/trace.log | tail -1 > $out/trace_bench_line.json
python3 tools/trace_summary.py /tmp/trace_$tag 8 60 > $out/fp32_train_kernels.txt
RCF_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/trace_b16_$tag -o r -- python3 bench.py --dtype bf16 --graph 0 --steps 5 --warmup 3 --preheat-s 0 --no-cpu-baseline > $out/trace_b16.log 2>&1
python3 tools/trace_summary.py /tmp/trace_b16_$tag 8 60 > $out/bf16_train_kernels.txt
RCF_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d /tmp/trace_3p_$tag -o r -- python3 bench.py --dtype f32_3plane --graph 0 --steps 5 --warmup 3 --preheat-s 0 --no-cpu-baseline > $out/trace_3p.log 2>&1
python3 tools/trace_summary.py /tmp/trace_3p_$tag 8 60 > $out/f32_3plane_train_kernels.txt
rocprofv3 --kernel-trace --stats -d /tmp/trace_inf_$tag -o r -- python3 bench.py --workload infer --graph 0 --steps 5 --warmup 3 --preheat-s 0 > $out/trace_inf.log 2>&1
python3 tools/trace_summary.py /tmp/trace_inf_$tag 8 40 > $out/bf16_infer_kernels.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=$out/pmc/$(echo $set | cut -d' ' -f1)
  # RCF_BATCH_PACK=0: the warm-up step and the counted step then issue the same dispatches (make_profile.py takes the second half)
  RCF_SINGLE_STREAM=1 RCF_BATCH_PACK=0 rocprofv3 --pmc $set --output-format csv -d $d -o b -- python3 bench.py --graph 0 --steps 1 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-side-leg --no-other-configs > $d.log 2>&1
done
python3 tools/make_profile.py /tmp/trace_$tag $out/pmc $out/trace_bench_line.json $tag $head > $out/make_profile.log 2>&1
# the bf16 configurations (BASELINE configs 2-4): the same three PMC passes per workload -> profiles/<tag>_pmc_<workload>.json, which their
# bench lines quote (traffic, MFMA-busy) when the kernel sources match
for wl in "bf16_train:--dtype bf16" "bf16_infer:--workload infer" "bf16_radarnet:--workload radarnet"; do
  name=${wl%%:*}; flags=${wl#*:}
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    d=$out/pmc_$name/$(echo $set | cut -d' ' -f1)
    mkdir -p $out/pmc_$name
    RCF_SINGLE_STREAM=1 RCF_BATCH_PACK=0 rocprofv3 --pmc $set --output-format csv -d $d -o b -- python3 bench.py $flags --graph 0 --steps 1 --warmup 1 --preheat-s 0 --no-cpu-baseline > $d.log 2>&1
  done
  python3 tools/pmc_families.py $out/pmc_$name $tag $name $head > $out/pmc_$name.log 2>&1
done
# every convolution launch of one eager step with its kernel id and duration (which layer runs where)
RCF_SINGLE_STREAM=1 RCF_DTYPE=fp32 python3 tools/layer_times.py 2>/dev/null | grep -v amdgpu > $out/fp32_train_layers.txt
RCF_SINGLE_STREAM=1 RCF_DTYPE=bf16 python3 tools/layer_times.py 2>/dev/null | grep -v amdgpu > $out/bf16_train_layers.txt
mkdir -p $out/profiles && cp profiles/${tag}_* $out/profiles/ 2>/dev/null
cat $out/pmc_bf16_*.log 2>/dev/null | grep -v amdgpu
tail -2 $out/make_profile.log
for f in bench_line bench_f32_3plane bench_bf16 bench_infer bench_radarnet bench_infer_f32 bench_radarnet_f32; do tail -1 $out/$f.log | cut -c1-260; done
