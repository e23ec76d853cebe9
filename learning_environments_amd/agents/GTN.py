"""agents.GTN (reference agents/GTN.py:1-76): re-exports Master/Worker and the launchers.

`run_gtn_on_single_pc(config)` in the reference forks one master + num_workers worker processes that talk through
files; here the master evaluates the whole population on the GPU in-process, so the launcher just runs it (under
torchrun it is SPMD: one process per GPU)."""
from .GTN_master import GTN_Master
from .GTN_worker import GTN_Worker


def _run_master(config):
    master = GTN_Master(config)
    master.clean_working_dir()
    return master.run()


def run_gtn_on_single_pc(config):
    """reference agents/GTN.py:14-45 forks a master and num_workers worker processes; here the master evaluates the population itself."""
    return _run_master(config)


def run_gtn_on_multiple_pcs(config, id):
    """reference agents/GTN.py:47-56: id -1 = the master of a file-transport run, id >= 0 = worker `id` (serves until the master's quit_flag)."""
    id = int(id)
    if id < -1:
        raise ValueError("Invalid ID")
    return _run_master(config) if id == -1 else GTN_Worker(id).run()
