#!/bin/bash
export PYTHONPATH=$PWD:$PWD/crnn-ocr-lite_amd:$PWD/tests
timeout 1500 python -m pytest tests/ -q -m gpu -x --tb=short -p no:cacheprovider -n 4 2>&1 | tail -4
