#!/usr/bin/env python3
"""Runs cases of the random campaigns (tests/campaign.py) through the reference's own build and the CPU restatement, each in a
child process (tests/campaign_child.py), and prints one line per case: identical / reference asserts / DIFFERENT.
usage: python profiles/reference_campaign.py scene|frame <first seed> <last seed> [workers]"""
import json, os, signal, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 2


def case(seed):
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "campaign_child.py"), kind, str(seed)], cwd=ROOT, capture_output=True, text=True)
    lines = [json.loads(line) for line in done.stdout.splitlines() if line.startswith("{")]
    end = lines[-1] if lines else {}
    if done.returncode == -signal.SIGABRT:
        assertion = [line for line in done.stderr.splitlines() if "Assertion" in line]
        return seed, "reference asserts", (assertion[-1][-120:] if assertion else "")
    if done.returncode == 0 and end.get("stage") == "done" and end["hits_equal"] and end["buffers_equal"]:
        return seed, "identical", f"{end['pixels']} pixels, {end['triangle_hits']} triangle and {end['sphere_hits']} sphere hits at the first vertex"
    return seed, "DIFFERENT", f"exit {done.returncode} {end} {done.stderr[-300:]}"


tally = {}
with ThreadPoolExecutor(workers) as pool:
    for seed, outcome, note in pool.map(case, range(first, last + 1)):
        tally[outcome] = tally.get(outcome, 0) + 1
        print(f"{kind} {seed}: {outcome}  {note}", flush=True)
print(f"{kind} seeds {first}..{last}: {tally}")
