out=$1; shift
: > $out
for v in "$@"; do
  export ORION_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/liborion_hip_$v.so
  echo "== $v" >> $out
  RAYS_MODES=0 timeout -k 10 200 python -u tools/bench_rays.py nanosuit soup 2>&1 | grep -E "auto" >> $out || exit 1
done
