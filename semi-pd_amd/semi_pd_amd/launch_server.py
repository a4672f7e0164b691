"""`python -m semi_pd_amd.launch_server --model-path <dir> --enable-semi-pd ...` — the reference's
`python -m sglang.launch_server` (python/sglang/launch_server.py) for the Semi-PD path."""
import argparse
import logging
import multiprocessing as mp
import sys


def main(argv=None):
    from semi_pd_amd.server_args import add_cli_args, from_cli_args
    parser = argparse.ArgumentParser(prog="semi_pd_amd.launch_server")
    add_cli_args(parser)
    args = parser.parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level.upper(), logging.INFO),
                        format="[%(asctime)s] %(message)s")
    server_args = from_cli_args(args)
    from semi_pd_amd.entrypoints.http_server import launch_server
    launch_server(server_args)


if __name__ == "__main__":
    mp.set_start_method("spawn", force=True)
    main(sys.argv[1:])
