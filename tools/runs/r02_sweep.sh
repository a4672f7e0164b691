# Poisson rate sweep, 512 requests per point, both fixed shapes; Semi-PD (default policy and the
# static 50/50 split) and the unified engine on the long shape
set -x
O=gpurun_out/r02_sweep; mkdir -p $O
COMMON="--num-requests 512 --no-cpu-baseline --no-static-split-wave --no-saturation-wave --no-kernel-timing --warmup 1 --steps 1"
R=4,8,12,16,24,32,48
python bench.py $COMMON --input-len 1024 --output-len 256 --request-rate 16 --rate-sweep 2,$R > $O/semi_pd_shared_in1024_out256.json 2> $O/a.err
python bench.py $COMMON --input-len 1024 --output-len 256 --request-rate 16 --rate-sweep $R --prefill-cu 50 --decode-cu 50 > $O/semi_pd_split50_in1024_out256.json 2> $O/b.err
python bench.py $COMMON --input-len 1024 --output-len 256 --request-rate 16 --rate-sweep $R --mode unified > $O/unified_in1024_out256.json 2> $O/c.err
python bench.py $COMMON --input-len 128 --output-len 64 --request-rate 64 --rate-sweep 4,8,16,32,64,128,256 > $O/semi_pd_shared_in128_out64.json 2> $O/d.err
python bench.py $COMMON --input-len 128 --output-len 64 --request-rate 64 --rate-sweep 4,8,16,32,64,128,256 --mode unified > $O/unified_in128_out64.json 2> $O/e.err
tail -c 600 $O/*.err
