import json,sys
for line in open(sys.argv[1]):
    if line.startswith('{"metric"'):
        l = json.loads(line)
        print(l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l["roofline"]["frac"], l["roofline"].get("traffic_stale"), l["roofline_fp64"]["frac"], l["roofline_fp64"].get("stale"), l["parity"]["full_batch_vs_exhaustive_bit_exact"], l["cpu_baseline"]["value"])
