"""The fit half of bench.py on its own (profiling runs, A/B of tunings): 1M x 1M, d = 128, WMRB, 20 interactions + 100 samples/user.
python scripts/fit_only.py [epochs] [tuning=value ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if __name__ == "__main__":
    args = sys.argv[1:]
    epochs = int(args[0]) if args and "=" not in args[0] else 2
    tunes = [a for a in args if "=" in a]
    if tunes:
        from tensorrec_amd import _native as N
        for t in tunes:
            name, value = t.split("=")
            N.set_tuning(name, int(value))
    print(json.dumps(bench.fit_epochs_per_sec(1_000_000, 1_000_000, 128, epochs=epochs)))
