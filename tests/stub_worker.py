"""A worker with no model: the executor's contract (``init_device`` ...
``execute_model`` by name through ``run_method``) without tensors or process
groups.  Every sequence's next token is ``(last token + rank of the replying
worker + 1) % 1000`` so a test can tell which rank answered."""
import os
import time

from vllm_distributed_amd.engine.types import KVCacheSpec, ModelRunnerOutput


class StubWorker:
    def __init__(self, vllm_config, local_rank, rank, distributed_init_method, is_driver_worker=False):
        self.vllm_config, self.local_rank, self.rank = vllm_config, local_rank, rank
        self.init_method = distributed_init_method
        self.is_driver_worker = is_driver_worker
        self.last = {}
        self.steps = 0

    def init_device(self):
        assert self.init_method and self.init_method.startswith("tcp://")

    def load_model(self):
        pass

    def get_kv_cache_spec(self):
        return KVCacheSpec(num_layers=1, num_kv_heads=1, head_dim=8, block_size=16, dtype_bytes=2)

    def determine_available_memory(self):
        return 1 << 20

    def initialize_from_config(self, cfg):
        self.num_blocks = cfg.num_blocks

    def compile_or_warm_up_model(self):
        pass

    def check_health(self):
        return True

    def whoami(self):
        return dict(rank=self.rank, local_rank=self.local_rank, pid=os.getpid(), driver=self.is_driver_worker,
                    host_ip=os.environ.get("VLLM_HOST_IP"))

    def nap(self, seconds):
        time.sleep(seconds)
        return self.rank

    def execute_model(self, so):
        self.steps += 1
        for nr in so.scheduled_new_reqs:
            self.last[nr.req_id] = nr.prompt_token_ids[-1]
        for rid in so.finished_req_ids:
            self.last.pop(rid, None)
        ids, toks = [], []
        for rid in so.num_scheduled_tokens:
            self.last[rid] = (self.last[rid] + self.rank + 1) % 1000
            ids.append(rid)
            toks.append([self.last[rid]])
        return ModelRunnerOutput(step_id=so.step_id, req_ids=ids, sampled_token_ids=toks)
