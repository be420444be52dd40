# Builder tool (GPU box): same-box A/B of when k_layernorm_ch requests gamma / beta / residual (LN_EARLY_MAX = 32: the build, ops.hip's default; 24: round 5's first setting; with the values for every
# instance; 0: behind the reductions as in rounds 1-4): single-utterance latency and the LayerNorm rows of its kernel stats.  The variants are whole libraries
# built beforehand into build/libsbv2_hip_ln{32,0}.so (see the round-5 notes in DESIGN.md); the product library is restored at the end.
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp $R/sbv2-api_amd/libsbv2_hip.so /tmp/lib_product.so
for v in product ln32 ln0 product; do
  [ $v = product ] && cp /tmp/lib_product.so $R/sbv2-api_amd/libsbv2_hip.so || cp $R/build/libsbv2_hip_$v.so $R/sbv2-api_amd/libsbv2_hip.so
  echo "== $v: $(python3 tools/b1_latency.py 40 2>/dev/null | head -1)"
  rm -rf /tmp/ln_ab; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ln_ab -- python3 tools/b1_latency.py 40 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ln_ab/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'layernorm' in r['Name']:
        print('   ', r['Name'][:48], 'calls/45', int(r['Calls']) / 45, 'avg us', round(float(r['AverageNs']) / 1e3, 2))
PY
done
cp /tmp/lib_product.so $R/sbv2-api_amd/libsbv2_hip.so
