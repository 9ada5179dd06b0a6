run() { python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-render-forward 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4))"; }
run base
for v in 256 1024; do DNS_MLP_FWD_BLOCKS=$v run fwd$v; done
for v in 128 512; do DNS_MLP_BWD_BLOCKS=$v run bwd$v; done
for v in 256 768 1024; do DNS_GEMM_BLOCKS=$v run gemm$v; done
run base2
