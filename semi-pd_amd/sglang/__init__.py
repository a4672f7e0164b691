"""Command-line alias only: lets `python -m sglang.launch_server --enable-semi-pd ...` (the reference's
launch command, docs + evaluation/benchmark_*_semi_pd.sh) start this engine when semi-pd_amd/ is on
PYTHONPATH instead of the reference.  Nothing else of the sglang namespace is provided."""
