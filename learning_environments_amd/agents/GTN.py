"""agents.GTN (reference agents/GTN.py:1-76): re-exports Master/Worker and the launchers.

`run_gtn_on_single_pc(config)` in the reference forks one master + num_workers worker processes that talk through
files; here the master evaluates the whole population on the GPU in-process, so the launcher just runs it (under
torchrun it is SPMD: one process per GPU)."""
from .GTN_master import GTN_Master
from .GTN_worker import GTN_Worker


def run_gtn_on_single_pc(config):
    gtn = GTN_Master(config)
    gtn.clean_working_dir()
    return gtn.run()


def run_gtn_on_multiple_pcs(config, id):
    if id == -1:
        gtn_master = GTN_Master(config)
        gtn_master.clean_working_dir()
        return gtn_master.run()
    elif id >= 0:
        gtn_worker = GTN_Worker(id)
        return gtn_worker.run()
    else:
        raise ValueError("Invalid ID")
