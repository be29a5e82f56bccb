p() { python -c "import sys, json; r = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['value'], r['ms_per_step'], r['config'].get('host_enqueue_ms_per_step'))"; }
echo "== graph + side, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
RCF_WGRAD_SIDE_STREAM=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | p
echo "== graph, no side, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
RCF_WGRAD_SIDE_STREAM=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | p
echo "== eager + side fp32, 100 steps"
RCF_WGRAD_SIDE_STREAM=1 python bench.py --graph 0 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | p
echo "== eager no side fp32, 100 steps"
RCF_WGRAD_SIDE_STREAM=0 python bench.py --graph 0 --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | p
echo "== graph no side fp32, 100 steps"
RCF_WGRAD_SIDE_STREAM=0 python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | p
echo "== eager + side bf16"
RCF_WGRAD_SIDE_STREAM=1 python bench.py --dtype bf16 --graph 0 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | p
echo "== eager no side bf16"
RCF_WGRAD_SIDE_STREAM=0 python bench.py --dtype bf16 --graph 0 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | p
