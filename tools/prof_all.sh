# final profiles of the round: headline bench (+ rocprof stats), pyramid3, postprocess  (through gpurun)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/prof_epi.sh
bash tools/prof_secondary.sh
