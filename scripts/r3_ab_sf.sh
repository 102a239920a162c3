#!/bin/bash
for i in 1 2; do bash scripts/ab_variants.sh "sf default" "1080p_jacobi1000 4k_jacobi1000 960x540_jacobi125" --no-estimate; done
