#!/bin/bash
# After tools/profile_round.sh <tag> <batch> on the GPU box (its CSVs merged back under gpurun_out/): copy them into profiles/
# under their round names and fold them into the per-kernel tables.   tools/fold_profiles.sh r03 512
set -e
TAG=${1:-r03}; B=${2:-512}
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out; P=$R/profiles
one() { find "$G/${TAG}_$1" -name "*$2" | head -1; }
cp "$(one stats kernel_stats.csv)" $P/${TAG}_kernel_stats_b$B.csv
cp "$(one pmc_fetch counter_collection.csv)" $P/${TAG}_pmc_fetch_size_b$B.csv
cp "$(one pmc_write counter_collection.csv)" $P/${TAG}_pmc_write_size_b$B.csv
cp "$(one pmc_sq1 counter_collection.csv)" $P/${TAG}_pmc_sq_pass1_b$B.csv
cp "$(one pmc_sq2 counter_collection.csv)" $P/${TAG}_pmc_sq_pass2_b$B.csv
python3 $R/tools/valu_mix.py $TAG > $P/${TAG}_valu_mix.json
python3 $R/tools/pmc_traffic.py $P/${TAG}_pmc_fetch_size_b$B.csv $P/${TAG}_pmc_write_size_b$B.csv $B $P/${TAG}_pmc_traffic_b$B.json $P/${TAG}_pmc_sq_pass1_b$B.csv $P/${TAG}_valu_mix.json > /dev/null
[ -f $P/${TAG}_bench_line.json ] && python3 $R/tools/kernel_roofline.py $TAG $B > $P/${TAG}_kernel_roofline.md
ls -la $P/${TAG}_*
