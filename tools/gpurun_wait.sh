#!/bin/bash
# Build-container helper: call gpurun, and when the pod has no free GPU slot (exit code 3: nothing ran, nothing was charged)
# try again every minute, for at most $TRIES attempts.  Any other outcome -- success or failure ON a box -- is returned as it is:
# a GPU command is never re-run by this script.
TRIES=${TRIES:-40}
for i in $(seq 1 "$TRIES"); do
    /usr/local/graft/bin/gpurun "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 60
done
exit 3
