"""Start ONE HIP GTN_Worker from the command line (reference experiments/GTN_Worker.py:13-16, what scripts/GTN_Worker:26-29 runs
once per SLURM array task):

    python -m learning_environments_amd.experiments.GTN_Worker <bohb_id> <id> [--seed N]

The worker waits in ./results/GTN_sync (relative to the current directory, as in the reference) for `<bohb_id>_<id>_input.pt`, runs the
1 + 2 * num_grad_evals inner loops of every evaluation as one launch of the fused kernel and answers with `<bohb_id>_<id>_result.pt`
until the master's payload carries quit_flag.  There is no CPU path: without a HIP device the first evaluation raises."""
import argparse
import sys


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="python -m learning_environments_amd.experiments.GTN_Worker", description=__doc__.split("\n\n")[0])
    ap.add_argument("bohb_id", type=int, help="id of the BOHB run the master belongs to (first part of the sync-file names)")
    ap.add_argument("id", type=int, help="worker id, 0 <= id < num_workers")
    ap.add_argument("--seed", type=int, default=None, help="fixed RNG seed (default: the reference's time-based seed)")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(sys.argv[1:] if argv is None else argv)
    for value in (args.bohb_id, args.id):          # the reference echoes its arguments
        print(value)
    if args.id < 0:
        raise ValueError("Invalid ID")
    import torch
    torch.set_num_threads(1)                        # reference :8
    from ..agents.GTN import GTN_Worker
    worker = GTN_Worker(id=args.id, bohb_id=args.bohb_id, seed=args.seed)
    worker.run()
    return 0


if __name__ == "__main__":
    sys.exit(main())
