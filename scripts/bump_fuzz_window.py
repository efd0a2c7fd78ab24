"""Move the live-reference fuzz window (tests/fuzz_window.json) for a new round: base += 1 000 000, round += 1.
Seeds that failed in the round being closed are appended to `regression` BY HAND with the reason (they keep their
arm and draw).  Usage: python scripts/bump_fuzz_window.py [--round N]"""
import argparse
import json
import os

PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "fuzz_window.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=None)
    a = ap.parse_args()
    with open(PATH) as fh:
        w = json.load(fh)
    w["round"] = a.round if a.round is not None else w["round"] + 1
    w["base"] = w["round"] * 1000000
    with open(PATH, "w") as fh:
        json.dump(w, fh, indent=1)
        fh.write("\n")
    print("window:", w["round"], w["base"], w["counts"])


if __name__ == "__main__":
    main()
