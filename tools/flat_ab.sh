# A/B of flat-kernel library variants on one box: bash tools/flat_ab.sh name1 name2 ...   ("" = the in-tree library)
# per variant: two flat_check runs (time, agreement with the general kernel), then the VALU count of one launch (PMC)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
V=$PWD/ndt_feature_graph_amd/variants
for v in "$@"; do
  L=""; [ "$v" != "tree" ] && L=$V/libndtgpu_$v.so
  for i in 1 2; do echo -n "$v: "; NDTGPU_LIB=$L timeout 120 python tools/flat_check.py 2>&1 | grep -E "^flat [0-9]|sampled maps" | tr '\n' ' '; echo; done
  NDTGPU_LIB=$L timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-include-regex "ndt_build_flat" --output-format csv -d /tmp/p_$v -o x -- python tools/flat_check.py > /tmp/l_$v.log 2>&1
  python tools/pmc_quick.py /tmp/p_$v flat | cut -c60-
done
