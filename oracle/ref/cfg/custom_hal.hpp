/* No custom HAL in the oracle build: every cv_hal_* hook stays the
 * NOT_IMPLEMENTED stub so the stock CPU path runs. */
#ifndef _CUSTOM_HAL_INCLUDED_
#define _CUSTOM_HAL_INCLUDED_
#endif
