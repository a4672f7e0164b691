"""Test plugin (ServerArgs.test_plugin): makes the capture of the small decode graphs fail inside every scheduler process
that loads it -- a synchronising call invalidates a stream capture, like a collective that refuses to be captured."""
from semi_pd_amd.model_executor import hip_graph_runner


def _fail_small_buckets(bs, out):
    if bs <= 4:
        out[1].sum().item()


hip_graph_runner.capture_fault_hook = _fail_small_buckets
