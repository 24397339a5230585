#!/bin/bash
# PMC detail for the round's new / dominant kernels (separate passes, --kernel-trace only)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for K in "cnx_mlp_kernel" "mit_mlp_kernel" "igemm_sbh_kernel<16, 16, 64" "dwconv3x3_gelu_mc"; do
  echo "=== $K"; PMC_KERNEL="$K" timeout 500 bash scripts/gpu_pmc_kernel.sh 2>&1 | grep -v "^W2026\|tool finalization" | tail -12
done
