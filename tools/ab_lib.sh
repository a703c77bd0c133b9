# A/B two builds of the library on the same box, interleaved: tools/ab_lib.sh <libA> <libB> <step_time args...>
A=$1; B=$2; shift 2
for rep in 1 2; do
  for lib in $A $B; do
    echo "== $lib"; BRIEF_LIB=$PWD/brief_pytorch_amd/$lib python3 tools/step_time.py "$@" 2>&1 | grep -v amdgpu.ids
  done
done
