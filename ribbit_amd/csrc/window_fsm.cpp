// window_fsm.cpp -- the state machine is header-only (window_fsm.h): it runs once per device event.
#include "window_fsm.h"
