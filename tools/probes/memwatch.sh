#!/bin/bash
# usage: memwatch.sh <cmd...>: prints vram/gtt used of every card every 2 s while the command runs
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
  line=""
  for c in /sys/class/drm/card*/device; do
    v=$(cat $c/mem_info_vram_used 2>/dev/null); g=$(cat $c/mem_info_gtt_used 2>/dev/null)
    [ -n "$v" ] && line="$line $(basename $(dirname $c)):vram=$((v>>20))MB,gtt=$((g>>20))MB"
  done
  echo "$(date +%s) $line rss=$(ps -o rss= -p $(pgrep -P $pid | head -1) 2>/dev/null)"
  sleep 2
done
wait $pid
