"""Prometheus metrics exposed on ``/metrics`` (names follow the ``vllm:``
convention so existing dashboards keep working)."""
from __future__ import annotations

import time
from typing import Dict, Optional

from prometheus_client import CollectorRegistry, Counter, Gauge, Histogram, generate_latest

_TTFT_BUCKETS = (0.001, 0.005, 0.01, 0.02, 0.04, 0.06, 0.08, 0.1, 0.25, 0.5, 0.75, 1, 2.5, 5, 10, 30)
_ITL_BUCKETS = (0.001, 0.0025, 0.005, 0.01, 0.02, 0.03, 0.05, 0.075, 0.1, 0.15, 0.2, 0.3, 0.5, 1, 2.5)
_E2E_BUCKETS = (0.1, 0.3, 0.5, 1, 1.5, 2.5, 5, 10, 20, 30, 60, 120, 300)


class EngineMetrics:
    def __init__(self, model_name: str):
        self.registry = CollectorRegistry()
        lbl = ["model_name"]
        self._labels = {"model_name": model_name}
        r = self.registry
        self.num_running = Gauge("vllm:num_requests_running", "Requests in the running batch", lbl, registry=r)
        self.num_waiting = Gauge("vllm:num_requests_waiting", "Requests waiting to be scheduled", lbl, registry=r)
        self.kv_usage = Gauge("vllm:gpu_cache_usage_perc", "Fraction of KV blocks in use", lbl, registry=r)
        self.prefix_queries = Gauge("vllm:prefix_cache_queries", "Prefix-cache tokens queried", lbl, registry=r)
        self.prefix_hits = Gauge("vllm:prefix_cache_hits", "Prefix-cache tokens hit", lbl, registry=r)
        self.preemptions = Gauge("vllm:num_preemptions_total", "Preemptions so far", lbl, registry=r)
        self.prompt_tokens = Counter("vllm:prompt_tokens_total", "Prefill tokens processed", lbl, registry=r)
        self.generation_tokens = Counter("vllm:generation_tokens_total", "Tokens generated", lbl, registry=r)
        self.requests = Counter("vllm:request_success_total", "Finished requests", lbl + ["finished_reason"],
                                registry=r)
        self.arrived = Counter("vllm:request_arrived_total", "Requests received", lbl, registry=r)
        self.ttft = Histogram("vllm:time_to_first_token_seconds", "TTFT", lbl, buckets=_TTFT_BUCKETS, registry=r)
        self.itl = Histogram("vllm:time_per_output_token_seconds", "Inter-token latency", lbl,
                             buckets=_ITL_BUCKETS, registry=r)
        self.e2e = Histogram("vllm:e2e_request_latency_seconds", "End-to-end latency", lbl,
                             buckets=_E2E_BUCKETS, registry=r)
        self._last_token_time: Dict[str, float] = {}

    def request_arrived(self) -> None:
        self.arrived.labels(**self._labels).inc()

    def observe_output(self, out, now: Optional[float] = None) -> None:
        now = now or time.time()
        n_new = len(out.new_token_ids)
        lab = self._labels
        if n_new:
            self.generation_tokens.labels(**lab).inc(n_new)
            last = self._last_token_time.get(out.request_id)
            if last is None:
                self.prompt_tokens.labels(**lab).inc(len(out.prompt_token_ids))
                if out.metrics is not None:
                    self.ttft.labels(**lab).observe(max(0.0, now - out.metrics.arrival_time))
            else:
                self.itl.labels(**lab).observe(max(0.0, (now - last) / n_new))
            self._last_token_time[out.request_id] = now
        if out.finished:
            self._last_token_time.pop(out.request_id, None)
            reason = out.outputs[0].finish_reason or "stop"
            self.requests.labels(finished_reason=reason, **lab).inc()
            if out.metrics is not None:
                self.e2e.labels(**lab).observe(max(0.0, now - out.metrics.arrival_time))

    def observe_scheduler(self, stats) -> None:
        lab = self._labels
        self.num_running.labels(**lab).set(stats.num_running)
        self.num_waiting.labels(**lab).set(stats.num_waiting)
        self.kv_usage.labels(**lab).set(stats.kv_cache_usage)
        self.prefix_queries.labels(**lab).set(stats.prefix_cache_queries)
        self.prefix_hits.labels(**lab).set(stats.prefix_cache_hits)
        self.preemptions.labels(**lab).set(stats.num_preemptions)

    def render(self) -> bytes:
        return generate_latest(self.registry)
