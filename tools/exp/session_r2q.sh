#!/bin/bash
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/bench_firbank.py | cut -c1-330
