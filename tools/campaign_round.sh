# Random parity campaigns of a round on fresh seeds: narrow windows (3-iteration kernel), wide windows (one-iteration kernel),
# one stream forced (the default splits small batches over two), strict order, integer volumes, larger images.
# usage (through gpurun): bash tools/campaign_round.sh <seed> [scale]  -> gpurun_out/campaign_*.log (the last line of each is its summary)
s=${1:-5100}; k=${2:-1}
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python "$@" > gpurun_out/campaign_$name.log 2>&1; echo "$name: $(tail -1 gpurun_out/campaign_$name.log)"; }
run narrow tools/random_campaign.py $((250*k)) $s
run wide tools/random_campaign.py $((150*k)) $((s+1)) wide
FDN_SUB_BATCHES=1 run one_stream tools/random_campaign.py $((100*k)) $((s+4))
FDN_SUB_BATCHES=1 run wide_one_stream tools/random_campaign.py $((100*k)) $((s+2)) wide
run strict tools/random_campaign.py $((100*k)) $((s+5)) strict
run int tools/random_campaign_int.py $((100*k)) $((s+3))
run big tools/random_campaign_big.py $((30*k)) $((s+6))
