cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ_[A-Z_0-9]+|GRBM_[A-Z_]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+)" | sort -u | tr '\n' ' ' | head -c 6000
