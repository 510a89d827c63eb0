#!/usr/bin/env python3
"""CLI entry (same role and flags as /root/reference/main.py:8-13):

    python main.py --path <dataset dir> [--model_type mc] [--batch_size 4] [--num_epochs 20] ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py --path ...
"""
from consistent_depth_amd import parallel
from consistent_depth_amd.params import Video3dParamsParser
from consistent_depth_amd.process import DatasetProcessor

if __name__ == "__main__":
    parallel.init()
    params = Video3dParamsParser().parse()
    DatasetProcessor().process(params)
