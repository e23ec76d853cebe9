python -m pytest tests -m gpu -q -x -k "two_rank_hip or graph or gtn_master" 2>&1 | tail -8
