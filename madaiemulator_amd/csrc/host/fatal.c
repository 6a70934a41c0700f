/*
 * fatal.c -- how the host layer ends the process on an error.
 *
 * The reference ends with a message and exit(1) wherever it cannot go on (libEmu/maxmultimin.c:495,
 * libEmu/emulate-fns.c:282-285, libEmu/regression.c:159, libEmu/emulator.c:764).  Its process is a handful of pthreads doing
 * arithmetic; this one has dozens of host threads inside the HIP runtime (lock-step groups, component threads, the
 * reader / writer of interactive_mode).  exit() from one of them runs the atexit handlers and static destructors -- the
 * HIP runtime's among them -- beside the live siblings, and two threads that fail together enter exit() together:
 * undefined behaviour, in practice signal 11 where the reference gives status 1 (round 4: a failed graph capture in a
 * threaded search ended the CLI with returncode -11).
 *
 * So termination is single-entry and teardown-free: the first caller wins an atomic gate, says what it has to say,
 * tells the other ranks of a multi-process run (ranks.c drops a `failed` marker), flushes stdout / stderr and leaves with
 * _exit(status) -- no atexit handler, no destructor.  Callers that lose the gate sleep until the winner has left.
 */
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include "libemu.h"

static volatile int g_gate = 0;
static void (*g_hook)(int status) = NULL;

/* what ranks.c wants done before the process leaves on an error (at most one hook: the rendezvous marker) */
void gpemu_host_on_exit(void (*hook)(int status)) { g_hook = hook; }

static void enter_gate(void)
{
	if (__sync_lock_test_and_set(&g_gate, 1))
		for (;;) pause();                     /* somebody else is already ending the process */
}

static void leave(int status)
{
	if (status != 0 && g_hook) g_hook(status);
	fflush(stdout);
	fflush(stderr);
	_exit(status);
}

void gpemu_host_exit(int status)
{
	enter_gate();
	leave(status);
	for (;;) pause();                         /* (not reached: _exit does not return) */
}

void gpemu_host_fatal(const char *fmt, ...)
{
	enter_gate();
	va_list ap;
	va_start(ap, fmt);
	vfprintf(stderr, fmt, ap);
	va_end(ap);
	leave(EXIT_FAILURE);
	for (;;) pause();
}
