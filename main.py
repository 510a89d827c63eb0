#!/usr/bin/env python3
"""Command-line entry of the MI355X fine-tuning engine.

Accepts the reference's command lines (flags and defaults of /root/reference/params.py, entry /root/reference/main.py:8-13):

    python main.py --path <dataset dir> [--model_type mc] [--batch_size 4] [--num_epochs 20] ...

and runs data-parallel when started once per GPU:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --path <dataset dir> ...
"""
import sys


def main(argv=None) -> int:
    from consistent_depth_amd import parallel
    from consistent_depth_amd.params import Video3dParamsParser
    from consistent_depth_amd.process import DatasetProcessor

    rank, _local_rank, world = parallel.init()          # no-op for a single process
    options = Video3dParamsParser().parse(argv)
    if world > 1 and rank == 0:
        print(f"data parallel over {world} processes (one GPU each)")
    DatasetProcessor().process(options)
    return 0


if __name__ == "__main__":
    sys.exit(main())
