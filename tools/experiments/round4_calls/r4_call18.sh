#!/bin/bash
# round 4, call 18: attention maps for ragged batches, finite streaming contexts in the label-exact modes
set -u
timeout 1200 python -m pytest tests -m gpu -q -x -k "attention_maps or streaming or finite_contexts or exact_mode or empty_row" 2>&1 | tail -6
