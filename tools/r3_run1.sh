set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/r3_pytest1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3_pytest1.log
tail -5 gpurun_out/r3_pytest1.log
python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r3_bench1.err
