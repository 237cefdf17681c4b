#!/bin/bash
# build_variant.sh NAME "FLAGS": builds libndtgpu with extra compile flags into ndt_feature_graph_amd/variants/libndtgpu_NAME.so
# (A/B runs on one GPU box: NDTGPU_LIB=<that file>), then restores the shipped library.
set -e
cd "$(dirname "$0")/.."
mkdir -p ndt_feature_graph_amd/variants
NDTGPU_BUILD_FLAGS="$2" python -c "from ndt_feature_graph_amd import binding; binding.build_library(force=True)" > /dev/null
cp ndt_feature_graph_amd/libndtgpu.so ndt_feature_graph_amd/variants/libndtgpu_$1.so
python -c "from ndt_feature_graph_amd import binding; binding.build_library(force=True)" > /dev/null
echo "built variants/libndtgpu_$1.so ($2)"
