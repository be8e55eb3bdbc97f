for ns in 1 2 3; do for tg in 128 64 96; do
echo "streams=$ns target=$tg: $(AFD_WGRAD_STREAMS=$ns AFD_WGW_TARGET=$tg python bench.py --no-kernels --no-cpu-baseline --no-sample --no-graph --steps 40 2>&1 | grep 'train:' )"
done; done
for wb in 2 8; do echo "streams=2 target=64 batch=$wb: $(AFD_WGRAD_BATCH=$wb AFD_WGRAD_STREAMS=2 AFD_WGW_TARGET=64 python bench.py --no-kernels --no-cpu-baseline --no-sample --no-graph --steps 40 2>&1 | grep 'train:' )"; done
