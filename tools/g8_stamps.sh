#!/bin/bash
# kept for the old name: the measurement build is tools/measure_lib.sh (gemm8 stamps + attention stamps + the 16 x 16 A/B family)
exec "$(dirname "$0")/measure_lib.sh" "$@"
