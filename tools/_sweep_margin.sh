mkdir -p gpurun_out/r05r; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mr in "0.25 3" "0.5 3" "1.0 3" "0.25 6" "0.5 6" "2.0 3"; do set -- $mr
  python3 tools/rank_alone.py --worlds 8 --scenes orbit --lanes 1,2 --speculate 1 --margin $1 --radius $2 --frames 60 --out gpurun_out/r05r/m$1_r$2.json > /dev/null 2> gpurun_out/r05r/m$1_r$2.err
  echo "margin $1 radius $2"; grep predicted gpurun_out/r05r/m$1_r$2.err | cut -c1-200
done
