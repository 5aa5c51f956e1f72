for g in 2560 3072 4096 5120; do echo "slices64 grid $g: $(W3_RANK_GRID=$g timeout -k 10 100 python bench.py --no-cpu-baseline --steps 2 2>/dev/null | grep -o '"predict_ms": [0-9.]*')"; done
