set -e
python tools/ab_bench.py 24 window_bits=16,chunk=256 window_bits=16,chunk=274 window_bits=16,chunk=228 window_bits=16,chunk=342 window_bits=17,chunk=256 window_bits=17,chunk=214 window_bits=17,chunk=321
python tools/ab_bench.py 20 chunk=64 chunk=86 chunk=43 chunk=128 chunk=171
python tools/ab_bench.py 22 chunk=256 chunk=342 chunk=171 chunk=228
