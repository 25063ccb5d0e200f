# Round 3: a short campaign over the new kernels (screening records, relaxed sums) beyond the committed seeds
mkdir -p gpurun_out
{
python scripts/fuzz_campaign.py 700000 600
python scripts/fuzz_campaign.py 710000 600 hostile
python scripts/fuzz_campaign.py 720000 400 lists wrappers
python scripts/fuzz_campaign.py 730000 300 hostile lists wrappers
python scripts/fuzz_campaign.py 740000 200 camera
python scripts/fuzz_campaign.py 750000 40 big
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_fuzz_campaign.txt
