#!/bin/bash
python -m pytest tests/test_gpu_errors_and_filters.py -m gpu -x -q -k float 2>&1 | tail -30
