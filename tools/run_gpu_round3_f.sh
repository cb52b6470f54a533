set -o pipefail
cd $GRAFT_REPO_ROOT
for pw in 8 16 4 2 32; do
  echo "== KIRAG_AMD_PATCH_W=$pw"
  KIRAG_AMD_PATCH_W=$pw bash tools/shape_trace.sh 1024 128 4 2>&1 | grep -E "shape|k_proj"
done
