"""Start num_workers HIP GTN_Workers on this machine, one child process each (reference experiments/GTN_Worker_single_pc.py:12-30,
which forks 16):

    python -m learning_environments_amd.experiments.GTN_Worker_single_pc [num_workers] [--bohb-id N] [--seed N]

The sync directory is emptied first (reference :15-16), then every worker is `python -m learning_environments_amd.experiments.GTN_Worker
<bohb_id> <id>`.  This process never touches the GPU itself -- the workers are children, nothing is exec'ed over a process that holds a
device -- and exits with the first non-zero exit code of a worker (0 when all ended on the master's quit_flag).  The workers share the
GPU: a launch whose teams of workgroups cannot assemble next to another worker's kernel is repeated with one workgroup per chain
(agents/GTN_worker.py:_run_chains)."""
import argparse
import os
import subprocess
import sys
import time


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="python -m learning_environments_amd.experiments.GTN_Worker_single_pc", description=__doc__.split("\n\n")[0])
    ap.add_argument("num_workers", type=int, nargs="?", default=16, help="worker processes to start (reference: 16)")
    ap.add_argument("--bohb-id", type=int, default=0, help="bohb_id of the master these workers serve (reference: 0)")
    ap.add_argument("--seed", type=int, default=None, help="worker i gets seed N + i (default: time-based seeds)")
    return ap.parse_args(argv)


def worker_command(bohb_id, id, seed=None):
    cmd = [sys.executable, "-m", "learning_environments_amd.experiments.GTN_Worker", str(bohb_id), str(id)]
    if seed is not None:
        cmd += ["--seed", str(seed + id)]
    return cmd


def main(argv=None):
    args = parse_args(sys.argv[1:] if argv is None else argv)
    if args.num_workers < 1:
        raise ValueError("num_workers must be at least 1")
    from ..agents.GTN_base import GTN_Base
    GTN_Base(args.bohb_id).clean_working_dir()
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    procs = [subprocess.Popen(worker_command(args.bohb_id, i, args.seed), env=env) for i in range(args.num_workers)]
    rc = 0
    try:
        # poll ALL children: a worker that dies while worker 0 still waits for the master's input must end the launcher (and the
        # remaining workers, in `finally`) with its exit code, not block behind a sequential wait (bench.launch_ranks does the same)
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                code = p.poll()
                if code is not None:
                    live.remove(p)
                    rc = rc or code
            if live and rc == 0:
                time.sleep(0.2)
    finally:
        for p in procs:                    # a worker that is still running when this process is interrupted: end exactly that child
            if p.poll() is None:
                p.terminate()
    return rc


if __name__ == "__main__":
    sys.exit(main())
