"""A worker with no model and (almost) no per-request Python: what is left of a step is the control
plane (pickle, frames, thread hand-offs) and the head's own scheduling code.  Every sequence's next token
is ``(step + rank) % 1000``."""
import numpy as np

from vllm_distributed_amd.engine.types import KVCacheSpec, ModelRunnerOutput


class FastStubWorker:
    def __init__(self, vllm_config, local_rank, rank, distributed_init_method, is_driver_worker=False):
        self.rank = rank
        self.steps = 0

    def init_device(self):
        pass

    def load_model(self):
        pass

    def get_kv_cache_spec(self):
        return KVCacheSpec(num_layers=1, num_kv_heads=1, head_dim=8, block_size=64, dtype_bytes=2)

    def determine_available_memory(self):
        return 1 << 34

    def initialize_from_config(self, cfg):
        pass

    def compile_or_warm_up_model(self):
        pass

    def check_health(self):
        return True

    def synchronize(self):
        import time
        return time.time()

    def get_stats(self):
        return {"rank": self.rank, "steps": self.steps}

    def execute_model(self, so):
        self.steps += 1
        ids = list(so.num_scheduled_tokens)
        tok = (self.steps + self.rank) % 1000
        return ModelRunnerOutput(step_id=so.step_id, req_ids=ids, sampled_token_ids=[[tok]] * len(ids))
