#!/bin/bash
set -u
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
