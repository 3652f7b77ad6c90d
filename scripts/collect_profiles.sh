#!/bin/bash
# copy the round's artifacts from gpurun_out/ (scratch) to profiles/ (tracked).  Usage: bash scripts/collect_profiles.sh r03
TAG=${1:-r06}
cp gpurun_out/gputests_$TAG.log profiles/${TAG}_gputests.log
cp gpurun_out/parity_ledger_$TAG.json profiles/${TAG}_parity_ledger.json
for f in gpurun_out/graderrs_$TAG.*.json; do n=$(basename $f | sed "s/graderrs_$TAG\.//"); cp $f profiles/${TAG}_grad_errors_$n; done
for w in 1.4b_b8 unidisc_s_b64 1.4b_interleaved_l4608_b2 1.4b_interleaved_l4608_b1 1.4b_adaln_b8 1.4b_b8_16cus_held; do
  python3 - "$w" "$TAG" <<'PY'
import sys, json
w, tag = sys.argv[1], sys.argv[2]
s = open(f"gpurun_out/bench_{w}_{tag}.json").read()
json.dump(json.loads(s[s.index("{"):]), open(f"profiles/{tag}_bench_{w}.json", "w"), indent=1)
PY
done
cp gpurun_out/kernel_stats_$TAG.csv profiles/${TAG}_kernel_stats_1.4b_b8.csv
cp gpurun_out/kernel_stats_${TAG}e.csv profiles/${TAG}_kernel_stats_1.4b_interleaved_l4608_b2.csv
cp gpurun_out/kernel_stats_${TAG}s.csv profiles/${TAG}_kernel_stats_unidisc_s_b64.csv
cp gpurun_out/pmc_traffic_$TAG.json profiles/${TAG}_pmc_hbm_traffic_per_kernel.json
cp gpurun_out/pmc_mfma_$TAG.json profiles/${TAG}_pmc_mfma_util_per_kernel.json
[ -f gpurun_out/ddp_rccl_world1_check.json ] && cp gpurun_out/ddp_rccl_world1_check.json profiles/${TAG}_ddp_rccl_world1_check.json
ls -la profiles | grep $TAG
