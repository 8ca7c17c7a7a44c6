set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02_a/pytest.log
tail -3 gpurun_out/r02_a/pytest.log
timeout 120 tools/valu_rates > gpurun_out/r02_a/valu_rates.json 2> gpurun_out/r02_a/valu_rates.err
mv gpurun_out/r02_a/pytest.log gpurun_out/r02_a/valu_rates.json /tmp/ 
bash tools/profile_round.sh r02_a > /tmp/prof.log 2>&1
cp /tmp/pytest.log /tmp/valu_rates.json gpurun_out/r02_a/
tail -5 /tmp/prof.log
cat gpurun_out/r02_a/valu_rates.json
