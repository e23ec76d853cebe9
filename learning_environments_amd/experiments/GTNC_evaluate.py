"""BOHB-compatible experiment wrapper around GTN_Master.

Same class / method names and return payload as the reference's experiment drivers
(experiments/GTNC_evaluate_cartpole.py:16-73) so an hpbandster worker -- which stays out of scope -- can call
`compute` unchanged: loss = number of NES generations needed (len(score_list)), +Inf on any failure; info carries the
error text and the per-generation score list.  The default config is passed in (the reference reads a YAML from cwd)."""
import traceback
from copy import deepcopy

from ..agents.GTN import GTN_Master


class ExperimentWrapper(object):
    def __init__(self, default_config):
        self.default_config = default_config

    def get_bohb_parameters(self):
        # experiments/GTNC_evaluate_cartpole.py:17-25
        return {'min_budget': 1, 'max_budget': 1, 'eta': 2, 'random_fraction': 1, 'iterations': 10000}

    def get_configspace(self):
        # :27-30 -- an empty ConfigSpace.ConfigurationSpace; the package is not part of this image
        try:
            import ConfigSpace as CS
        except ImportError as e:
            raise NotImplementedError("ConfigSpace is not installed; hpbandster integration is out of scope") from e
        return CS.ConfigurationSpace()

    def get_specific_config(self, cso, default_config, budget):
        # :32-34
        return deepcopy(default_config)

    def compute(self, working_dir, bohb_id, config_id, cso, budget, *args, **kwargs):
        # :36-73; the reference's bare `except` maps every failure to loss = +Inf
        config = self.get_specific_config(cso, self.default_config, budget)
        try:
            gtn = GTN_Master(config, bohb_id=bohb_id, bohb_working_dir=working_dir)
            _, score_list, _ = gtn.run()
            score = len(score_list)
            error = ""
        except Exception:
            score = float('Inf')
            score_list = []
            error = traceback.format_exc()
        return {"loss": score, "info": {'error': str(error), 'score_list': str(score_list)}}
