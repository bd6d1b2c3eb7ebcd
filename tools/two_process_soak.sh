#!/bin/bash
# Two PROCESSES on one GPU at once, each launching relay-segmented chain kernels (tools/relay_soak.py: ragged chain counts, random step
# counts, every comparison bit for bit against the unsegmented launch) while a third fits VI with cooperative launches in a loop
# (tools/vi_soak.py): the ticket-ordered relay and the cooperative VI launch must neither dead-lock nor time out when workgroups of
# different processes interleave on the device (overlapping relay launches; VERDICT r05: "two processes on one GPU").
# usage: tools/two_process_soak.sh [seconds]
R="${GRAFT_REPO_ROOT:-$(pwd)}"; T="${1:-60}"; O="$R/gpurun_out/two_process_soak"; mkdir -p "$O"
timeout $((T + 240)) python3 "$R/tools/relay_soak.py" 11 "$T" > "$O/relay_a.txt" 2>&1 & A=$!
timeout $((T + 240)) python3 "$R/tools/relay_soak.py" 12 "$T" > "$O/relay_b.txt" 2>&1 & B=$!
timeout $((T + 240)) python3 "$R/tools/vi_soak.py" 5 "$T" > "$O/vi.txt" 2>&1 & C=$!
wait $A; ra=$?; wait $B; rb=$?; wait $C; rc=$?
echo "relay soak A rc=$ra: $(grep -v DEBUG "$O/relay_a.txt" | tail -1)"
echo "relay soak B rc=$rb: $(grep -v DEBUG "$O/relay_b.txt" | tail -1)"
echo "vi soak      rc=$rc: $(grep -v DEBUG "$O/vi.txt" | tail -1)"
