"""Child process of test_one_rank_rccl_group_runs_the_collective_path (tests/test_distributed_gloo.py): two GTN_Master generations on
cuda:0, with a ONE-RANK RCCL process group (`--group 1`: the communicator is created before anything else touches the GPU, the fitness
records go through dist.all_gather_into_tensor on a float64 device tensor, the captured generation is the two graphs around it) or
without any process group (`--group 0`: one graph, no collective).  Writes theta, the score lists and the path counters to --out."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", type=int, required=True)
    ap.add_argument("--port", type=int, default=29631)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--workdir", required=True)
    args = ap.parse_args()
    os.chdir(args.workdir)
    import numpy as np
    import torch
    import torch.distributed as dist
    if args.group:
        # first GPU-touching call of the process: the communicator (no re-exec, no earlier HIP call)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % args.port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.cuda.set_device(0)
    from test_distributed_gloo import _small_config
    from learning_environments_amd.agents.GTN import GTN_Master
    torch.manual_seed(0)
    m = GTN_Master(_small_config(5), bohb_id=0, seed=11, graph=bool(args.graph))
    with torch.no_grad():
        m.synthetic_env_orig.env.done_net[-1].bias.fill_(-10.0)
    mean_score, mean_list, _ = m.run()
    torch.cuda.synchronize()
    np.savez(args.out, theta=m.theta.cpu().numpy(), score=np.array(m.score_list), score_orig=np.array(m.score_orig_list), mean=np.array(mean_list),
             has_group=int(m.has_group), collectives=int(m.collectives_run), use_graph=int(m.use_graph), replays=int(m.graph_replays),
             two_graphs=int(getattr(m, "_graph2", None) is not None), backend=str(dist.get_backend()) if args.group else "none",
             capture_error=str(getattr(m, "graph_capture_error", "")))
    if args.group:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    main()
