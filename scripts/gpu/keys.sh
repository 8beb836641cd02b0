#!/bin/bash
python -m pytest tests/test_gpu_groupby.py -m gpu -x -q -k "transformed_keys" 2>&1 | tail -30
