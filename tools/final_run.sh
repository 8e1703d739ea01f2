# End-of-round evidence on one box: 2-rank control flow (gloo, both ranks on cuda:0), the whole -m gpu suite, the default
# bench line, rocprofv3 kernel statistics + the two HBM counter passes.  Results: gpurun_out/final/ and gpurun_out/prof_<tag>/
set -u
tag=${1:-r02}
out=gpurun_out/final; mkdir -p $out
DP="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 3 --backend gloo --ranks-share-gpu0 --no-cpu-baseline --no-kernels-alone --steady-after 0 --reserve-gb 2"
timeout 600 $DP > $out/dp2_weak.json 2> $out/dp2_weak.err
timeout 600 $DP --scaling strong > $out/dp2_strong.json 2> $out/dp2_strong.err
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $out/full_suite.txt
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
bash tools/collect_profiles.sh $tag > $out/collect.log 2>&1
tail -2 $out/full_suite.txt; tail -c 400 $out/dp2_weak.json; echo; tail -c 300 $out/dp2_strong.json; echo; cut -c1-300 $out/bench.json
