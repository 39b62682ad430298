mkdir -p gpurun_out/r05t; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r05t/t.log; tail -5 gpurun_out/r05t/t.log
python3 tools/rank_alone.py --worlds 8 --scenes orbit --lanes 1,2 --speculate 1 --frames 60 --out gpurun_out/r05t/ra4.json > /dev/null 2> gpurun_out/r05t/ra4.err; grep predicted gpurun_out/r05t/ra4.err | cut -c1-200
python3 tools/rank_alone.py --workload cfg5 --worlds 8 --scenes orbit --lanes 1,2 --frames 50 --out gpurun_out/r05t/ra5.json > /dev/null 2> gpurun_out/r05t/ra5.err; grep predicted gpurun_out/r05t/ra5.err | cut -c1-200
