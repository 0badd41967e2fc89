set -e
bash tools/measure.sh r2_b
