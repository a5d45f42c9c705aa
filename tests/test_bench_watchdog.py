"""bench.py's phase markers and host watchdog (no GPU): a run that goes silent must say where, and end with a status."""
import json
import subprocess
import sys
import textwrap
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def run(code: str, timeout: float = 60.0):
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=REPO, capture_output=True, text=True, timeout=timeout)


def test_watchdog_reports_the_phase_and_the_engine_words_and_exits_3():
    """A stub 'stuck launch': no phase change for longer than the limit -> one line with the phase and the host-visible
    words (first error code, epoch of the last completed launch), exit status 3 through os._exit (never a re-exec)."""
    p = run("""
        import time, bench
        wd = bench.Watchdog(limit_s=1.0, tag="stub")
        wd.words = lambda: (0x41000007, 123)
        wd.phase("warm-up, 16 steps")
        time.sleep(30)
        print("not reached")
    """)
    assert p.returncode == 3, (p.returncode, p.stderr)
    assert "not reached" not in p.stdout
    assert "phase 'warm-up, 16 steps'" in p.stderr
    assert "WATCHDOG" in p.stderr and "first error 0x41000007" in p.stderr and "epoch 123" in p.stderr


def test_watchdog_is_quiet_while_phases_change_and_after_stop():
    p = run("""
        import time, bench
        wd = bench.Watchdog(limit_s=1.5, tag="stub")
        for i in range(4):
            wd.phase(f"step {i}")
            time.sleep(0.6)
        for i in range(4):
            wd.touch()
            time.sleep(0.6)
        wd.stop()
        time.sleep(2.5)
        print("done")
    """)
    assert p.returncode == 0 and "done" in p.stdout and "WATCHDOG" not in p.stderr, (p.returncode, p.stderr)
    assert p.stderr.count("phase '") == 4


def test_watchdog_survives_unreadable_words_and_can_be_switched_off():
    p = run("""
        import time, bench
        wd = bench.Watchdog(limit_s=0.5, tag="stub")
        def boom():
            raise RuntimeError("gone")
        wd.words = boom
        time.sleep(10)
    """)
    assert p.returncode == 3 and "unreadable (gone)" in p.stderr
    p = run("""
        import time, bench
        wd = bench.Watchdog(limit_s=0)
        time.sleep(1.0)
        print("alive")
    """)
    assert p.returncode == 0 and "alive" in p.stdout


def test_the_bench_line_carries_the_committed_full_cpu_run():
    """SURVEY 8(d)'s >= 8-token CPU figure travels in cpu_baseline.full_run with its source file."""
    sys.path.insert(0, str(REPO))
    import bench

    full = bench.committed_full_cpu_run("llama2-7b-int4")
    assert full is not None and full["source"].startswith("profiles/") and (REPO / full["source"].split(" ")[0]).exists()
    assert "8 single-token decode steps" in full["sample"] and 0 < full["value"] < 1 and full["kind"] == "port"
    assert bench.committed_full_cpu_run("no-such-workload") is None
