# Round 3: the longer campaign over the new default kernels (about 9 minutes of GPU time)
mkdir -p gpurun_out
{
python scripts/fuzz_campaign.py 800000 12000
python scripts/fuzz_campaign.py 820000 12000 hostile
python scripts/fuzz_campaign.py 840000 5000 lists wrappers
python scripts/fuzz_campaign.py 850000 4000 hostile lists wrappers
python scripts/fuzz_campaign.py 860000 3000 camera
python scripts/fuzz_campaign.py 870000 300 big
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_fuzz_campaign_big.txt
