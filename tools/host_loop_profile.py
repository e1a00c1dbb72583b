"""Developer tool (GPU box): cProfile of bench.py's host-inclusive loop -- where the host's share of a step goes.
usage: python3 tools/host_loop_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + ["--steps", sys.argv[1] if len(sys.argv) > 1 else "400", "--warmup", "10", "--no-cpu-baseline"]
import bench  # noqa: E402

orig = bench.host_inclusive_rate


def profiled(*a, **k):
    pr = cProfile.Profile()
    pr.enable()
    out = orig(*a, **k)
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stderr)
    st.sort_stats("cumulative").print_stats(28)
    st.sort_stats("tottime").print_stats(18)
    return out


bench.host_inclusive_rate = profiled
bench.main()
