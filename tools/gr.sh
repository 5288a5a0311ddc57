#!/bin/bash
# helper for the build container: gpurun with gpurun_out/r03 created first
exec gpurun --timeout ${T:-1200} -- "mkdir -p gpurun_out/r03; $*"
