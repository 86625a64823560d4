"""launch.spawn_ranks: the rank launcher behind `bench.py --gpus N` (no torchrun), on the CPU with gloo.

Two ranks rendezvous through the environment the launcher sets (env://, 127.0.0.1), all-reduce, and rank 0's stdout comes
back; a failing rank fails the launch and takes the surviving rank (parked in a collective) down with it."""
import importlib
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
launch = importlib.import_module("iclr2025_3d-mom_amd.launch")


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_two_ranks_rendezvous_and_rank0_stdout_is_returned(tmp_path):
    script = _script(tmp_path, """
        import os, torch, torch.distributed as dist
        dist.init_process_group("gloo")            # env:// : RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT from the launcher
        r, w = dist.get_rank(), dist.get_world_size()
        assert int(os.environ["LOCAL_RANK"]) == r and os.environ["MASTER_ADDR"] == "127.0.0.1"
        t = torch.tensor([float(r + 1)])
        dist.all_reduce(t)
        dist.barrier()
        print(f"rank {r} of {w}: sum {t.item():.0f}", flush=True)
        dist.destroy_process_group()
    """)
    rc, out = launch.spawn_ranks(2, [sys.executable, script], timeout=120)
    assert rc == 0
    lines = [l for l in out.strip().splitlines() if not l.startswith("[Gloo]")]     # gloo announces itself on stdout
    assert lines == ["rank 0 of 2: sum 3"]                  # rank 0's stdout only


def test_a_failing_rank_fails_the_launch_and_the_other_rank_is_stopped(tmp_path):
    script = _script(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)                                     # stands for a rank waiting in a collective for its dead peer
    """)
    rc, out = launch.spawn_ranks(2, [sys.executable, script], timeout=120)
    assert rc == 7 and out == ""


def test_rank_detection(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert not launch.launched_by_a_launcher()
    assert launch.main_or_spawn(1, "unused.py", []) is False     # one process wanted: the caller carries on
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert launch.launched_by_a_launcher()
    assert launch.main_or_spawn(2, "unused.py", []) is False     # already a rank
