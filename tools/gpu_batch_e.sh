#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_pca.py -x -q -m gpu -k "msdi or grid" 2>&1 | tail -15
