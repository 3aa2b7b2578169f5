#!/bin/bash
# builds the sequential-walk twin of the library next to the product and runs tools/walk_closed_form.py on both, interleaved
set -e
cd "$(dirname "$0")/.."
SEQ=vokselis_amd/_lib/libvokselis_hip_seq.so
if [ ! -f $SEQ ] || [ vokselis_amd/csrc/vk_kernels.hpp -nt $SEQ ]; then
  /opt/rocm/bin/hipcc $(python -c "import __graft_entry__ as G; print(' '.join(G.HIPCC_FLAGS))") -DVK_WALK_SEQUENTIAL -o $SEQ vokselis_amd/csrc/vk_api.hip
fi
for lib in $SEQ vokselis_amd/_lib/libvokselis_hip.so $SEQ vokselis_amd/_lib/libvokselis_hip.so; do
  timeout -k 10 300 python tools/walk_closed_form.py $lib "$@" 2>&1 | grep "^{"
done
