import json, sys
d = json.load(open(sys.argv[1]))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for k, v in list(d.items())[:int(sys.argv[3]) if len(sys.argv) > 3 else 50]:
    print(f"{v['ms_per_step']:7.2f} ms {v['us_per_launch']:8.1f} us x{v['launches']//steps:3d} {v['tflops']:6.1f} TF {v['gbs']:6.0f} GB/s  {k}")
