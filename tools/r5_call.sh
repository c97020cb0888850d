cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
timeout 1200 python -m pytest tests/test_lazy_flow_gpu.py tests/test_train_resume_gpu.py tests/test_train_gpu.py tests/test_model_api.py -m gpu -x -q > gpurun_out/c5/pytest.log 2>&1; tail -5 gpurun_out/c5/pytest.log
for i in 1 2; do
TEF_LAZY_FLOWS=0 timeout 600 python bench.py --mode dropin --steps 5 2>/dev/null | cut -c150-200
timeout 600 python bench.py --mode dropin --steps 5 2>/dev/null | cut -c150-200
done
