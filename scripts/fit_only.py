"""The fit half of bench.py on its own (profiling runs): 1M x 1M, d = 128, WMRB, 20 interactions + 100 samples/user."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if __name__ == "__main__":
    print(json.dumps(bench.fit_epochs_per_sec(1_000_000, 1_000_000, 128, epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 2)))
