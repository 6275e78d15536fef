# bench.py's N > 1 flow rehearsed on ONE GPU: W ranks (gloo for the votes and reductions, every rank on cuda:0), the C-ABI
# exchange pano_gather_slots between real peers through the RCCL test double (tests/src/fake_rccl.cpp).  bash tools/bench_rehearsal.sh [W...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "import sys; sys.path.insert(0, '$R/tests'); import helpers; print(helpers.build_fake_rccl())" > /tmp/fake_path.txt || exit 1
FAKE=$(tail -1 /tmp/fake_path.txt)
for W in ${@:-2 4}; do
  PANO_BENCH_BACKEND=gloo PANO_RCCL_LIB=$FAKE FAKE_RCCL_TIMEOUT_S=120 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $W \
    --master-addr 127.0.0.1 --master-port $((29610 + W)) $R/bench.py --gpus $W --steps 20 --warmup 5 > $R/gpurun_out/rehearsal_$W.json 2> $R/gpurun_out/rehearsal_$W.err || { tail -20 $R/gpurun_out/rehearsal_$W.err; exit 1; }
  tail -1 $R/gpurun_out/rehearsal_$W.json | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('W=$W', d['value'], d['ms_per_step'], d['from_idle'], d['multi_gpu'])"
done
