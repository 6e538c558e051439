#!/bin/bash
# A/B of the two-lane forms of the G2 point chains (NBLS_PT_LS2_MAX) and of the sign-aligned ladder (NBLS_G2_SAC_MAX): sign / getPublicKey against the batch size, and one verify
for v in "NBLS_PT_LS2_MAX=0 NBLS_G2_SAC_MAX=0" "NBLS_PT_LS2_MAX=0 NBLS_G2_SAC_MAX=6144" "NBLS_PT_LS2_MAX=4096 NBLS_G2_SAC_MAX=6144"; do
  echo "== $v"
  env $v python tools/sign_sizes.py 1,512,2048,4096 2>&1 | grep -v amdgpu.ids
  env $v python tools/verify_breakdown.py 1 2>&1 | grep -E "verify_batch_dev|h2c_c1|h2c_c2" 
done
