#!/bin/bash
# Build an A/B variant of the library into tools/bin/ (git-ignored; travels to the GPU box):
#   tools/variant.sh NAME "src1.hip:flags" ["src2.hip:flags" ...]
# Sources not named are linked from the objects of the last regular build (python -m waveverify_amd.build).
# Benchmarks take the variant with --lib tools/bin/libwv_NAME.so (they set waveverify_amd._lib.LIB_PATH before loading).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
objs=""
declare -A repl
for spec in "$@"; do
  src=${spec%%:*}; flags=${spec#*:}
  fp=fast; case $src in wv_kernels.hip|wv_k1.hip|wv_rb.hip|wv_h16.hip) fp=fast;; *) fp=off;; esac
  o=tools/bin/${name}_${src%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=$fp $flags -c waveverify_amd/csrc/$src -o $o
  repl[$src]=$o
done
for src in wv_kernels.hip wv_k1.hip wv_rb.hip wv_h16.hip wv_model.hip wv_ops.hip wv_train.hip wv_aug.hip wv_fx.hip; do
  if [ -n "${repl[$src]}" ]; then objs="$objs ${repl[$src]}"; else objs="$objs waveverify_amd/lib/${src%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libwv_$name.so $objs
python3 -c "import ctypes,sys; ctypes.CDLL(sys.argv[1])" tools/bin/libwv_$name.so   # every kernel stub resolves (hipcc has dropped some silently)
echo tools/bin/libwv_$name.so
