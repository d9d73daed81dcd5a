#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_pca.py -x -q -m gpu -k "beyond_512 or annular" 2>&1 | tail -8
