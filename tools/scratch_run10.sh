#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
o=gpurun_out/r6; mkdir -p $o
timeout 900 python tools/kernel_probe.py --which mix_fwd_add,mix_fwd --ldm 10 --iters 20 --sweep "mix_tickets=0;mix_tickets=8,mix_ticket_tile=4;mix_tickets=8,mix_ticket_tile=2;mix_tickets=8,mix_ticket_tile=8;mix_tickets=4,mix_ticket_tile=8;mix_tickets=4,mix_ticket_tile=16;mix_tickets=2,mix_ticket_tile=16;mix_tickets=2,mix_ticket_tile=32;mix_tickets=1,mix_ticket_tile=32;mix_tickets=1,mix_ticket_tile=64;mix_tickets=12,mix_ticket_tile=4;mix_tickets=0;mix_tickets=8,mix_ticket_tile=4" > $o/probe_tk2.txt 2>&1; cat $o/probe_tk2.txt
