# Ablation builds of skinny.hip (measurement only): what each of A loads / the six MFMA products / the C stores costs, and the
# effect of the workgroup size.  Builds variant libraries next to the real one and times the three config-2 layer-0 products on each.
#   (container)  bash tools/exp_skinny_ablate.sh build      (GPU box)  bash tools/exp_skinny_ablate.sh run
cd "$(dirname "$0")/.."
VARIANTS="base:-DSK_BASE mfma1:-DSK_ABLATE_MFMA nostore:-DSK_ABLATE_STORE noload:-DSK_ABLATE_LOAD w8:-DSK_WAVES=8 w16:-DSK_WAVES=16 nostore_noload:-DSK_ABLATE_STORE,-DSK_ABLATE_LOAD"
if [ "$1" = build ]; then
  mkdir -p bot_amd/lib/exp
  OBJS=$(ls bot_amd/lib/obj/*.o | grep -v skinny.o)
  for v in $VARIANTS; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr ',' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c bot_amd/csrc/skinny.hip -o /tmp/skinny_$name.o -Iinclude || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bot_amd/lib/exp/libbot_gnn_$name.so $OBJS /tmp/skinny_$name.o -L/opt/rocm/lib -lhipblaslt || exit 1
  done
  ls -la bot_amd/lib/exp
else
  for v in $VARIANTS; do
    name=${v%%:*}
    echo "== $name"
    BOT_AMD_LIB=$PWD/bot_amd/lib/exp/libbot_gnn_$name.so python tools/exp_skinny.py 2>&1 | grep -v amdgpu.ids
  done
fi
