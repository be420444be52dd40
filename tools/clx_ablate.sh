# Builder tool (build container): diagnostic builds of the library with conv_clx.hip's CLX_ABL ablations under build/abl_<n>/ (wrong results by design);
# on the GPU box tools/clx_ablate_run.sh times them beside the product build.    bash tools/clx_ablate.sh 1 2 3 4
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
for n in "$@"; do
  D=$R/build/abl_$n
  rm -rf $D && mkdir -p $D/tools
  cp -r $R/sbv2-api_amd $D/ && cp $R/sbv2_api_amd.py $D/ && cp $R/tools/clx_timeline.py $D/tools/ && cp -r $R/include $D/
  rm -f $D/sbv2-api_amd/csrc/conv_clx.o $D/sbv2-api_amd/libsbv2_hip.so
  make -C $D/sbv2-api_amd/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -DCLX_ABL=$n" ../libsbv2_hip.so 2>&1 | grep -E "error|warning: unused" || true
  ls -la $D/sbv2-api_amd/libsbv2_hip.so
done
