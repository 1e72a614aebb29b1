#!/bin/bash
# usage: tools/gdb_run.sh <log> <timeout_s> <cmd...> : run a command under rocgdb, native backtraces on a fatal signal
log=$1; shift; to=$1; shift
timeout $to /opt/rocm/bin/rocgdb -batch -q \
  -ex "set pagination off" -ex "set confirm off" -ex "set print thread-events off" \
  -ex "handle SIGSEGV stop print nopass" -ex "handle SIGPIPE nostop noprint pass" \
  -ex run -ex "echo \n==== BACKTRACE ====\n" -ex bt -ex "echo \n==== ALL THREADS ====\n" \
  -ex "thread apply all bt 25" -ex "info sharedlibrary" --args "$@" > $log 2>&1
echo "rc=$?" >> $log
