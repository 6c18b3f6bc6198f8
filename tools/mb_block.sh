set -e
mkdir -p gpurun_out/mb
{
for n in 2000 3000 4096 5000 6000 7000 8192 9216 10240 12288 14336 16384 18432 20480 24576; do
  for np in 2 3 4 5 6 7 8; do
    timeout -k 5 60 tools/microbench_block $n $np 1 1 1000
  done
done
timeout -k 5 60 tools/microbench_block 8192 8 1 0 1000
timeout -k 5 60 tools/microbench_block 8192 8 0 1 1000
timeout -k 5 60 tools/microbench_block 8192 8 0 0 1000
timeout -k 5 60 tools/microbench_block 16384 8 0 1 1000
timeout -k 5 60 tools/microbench_block 8192 8 1 1 1000 0.01
timeout -k 5 60 tools/microbench_block 8192 8 0 1 1000 0.01
timeout -k 5 60 tools/microbench_block 8200 8 1 1 1000
timeout -k 5 60 tools/microbench_block 8192 8 1 1 1000 0 5
timeout -k 5 60 tools/microbench_block 8200 5 0 1 1000 0 40
timeout -k 5 60 tools/microbench_block 3000 3 0 1 1000 0 40
} > gpurun_out/mb/block2.txt 2>&1
tail -5 gpurun_out/mb/block2.txt
