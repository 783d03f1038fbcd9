import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as b
print(json.dumps(b.c5()))
