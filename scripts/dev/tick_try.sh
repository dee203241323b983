set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
SFX_FUSE_TICK=0 timeout -k 10 300 python tests/tick_worker.py > gpurun_out/r03t/w0.json 2> gpurun_out/r03t/w0.err
echo staged done
SFX_FUSE_TICK=1 timeout -k 10 300 python tests/tick_worker.py > gpurun_out/r03t/w1.json 2> gpurun_out/r03t/w1.err
echo fused done
