/*
 * perseus-sdr.h -- drop-in C API of the MI355X-native Perseus DSP path.
 *
 * Same names, signatures, constants, error codes and call-order rules as the
 * single public header of libperseus-sdr (reference perseus-sdr.h:41-366), so
 * that a client written against the reference recompiles against this file
 * unchanged.  The text, macros and implementation are new; the USB/FX2/FPGA
 * plumbing of the reference is replaced by a synthetic/file source
 * (include/perseus-amd-ext.h) and the FPGA's NCO + decimating FIR chain runs on
 * the GPU (include/perseus_ddc.h).  Unlike the reference header this one does
 * not pull in <libusb-1.0/libusb.h>; it includes <stdint.h> and <limits.h>
 * itself (the reference relied on libusb.h for uint8_t / INT_MAX).
 *
 * Reference line numbers are cited per declaration.
 */
#ifndef PERSEUS_SDR_AMD_H
#define PERSEUS_SDR_AMD_H

#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

/* ---- constants (reference perseus-sdr.h:41-61) ----------------------------- */
#define PERSEUS_PRODCODE      0x8014
#define PERSEUS_ADC_CLK_FREQ  80000000
#define PERSEUS_DDC_FREQ_MIN  0
#define PERSEUS_DDC_FREQ_MAX  (PERSEUS_ADC_CLK_FREQ / 2)

#define PERSEUS_ATT_0DB   0
#define PERSEUS_ATT_10DB  1
#define PERSEUS_ATT_20DB  2
#define PERSEUS_ATT_30DB  3

/* ---- product id record, 12 bytes packed (reference perseus-sdr.h:68-75) ----- */
typedef struct __attribute__((packed, aligned(1))) {
    uint16_t sn;            /* serial number                 */
    uint16_t prodcode;      /* PERSEUS_PRODCODE              */
    uint8_t  hwrel;         /* hardware release              */
    uint8_t  hwver;         /* hardware version              */
    uint8_t  signature[6];  /* product signature             */
} eeprom_prodid;

/* ---- data callback (reference perseus-sdr.h:81); return value is ignored
 *      (reference perseus-in.c:207).  `buf` belongs to the library and is only
 *      valid during the call (reference perseus-in.c:263).                      */
typedef int (*perseus_input_callback)(void *buf, int buf_size, void *extra);

/* opaque descriptor; points into a static library table, never freed by the
 * client (reference perseus-sdr.h:83-84, perseus-sdr.c:43-47)                   */
struct perseus_descr_ds;
typedef struct perseus_descr_ds perseus_descr;

#ifdef __cplusplus
extern "C" {
#endif

void perseus_set_debug(int debug_level);                                   /* ref :99  */
int  perseus_init(void);                 /* returns the NUMBER of receivers   ref :106 */
int  perseus_exit(void);                                                   /* ref :113 */
perseus_descr *perseus_open(int nDev);   /* NULL on error                     ref :124 */
int  perseus_close(perseus_descr *descr);                                  /* ref :133 */
int  perseus_firmware_download(perseus_descr *descr, char *fname);         /* ref :145 */
int  perseus_get_product_id(perseus_descr *descr, eeprom_prodid *prodid);  /* ref :156 */
int  perseus_set_attenuator(perseus_descr *descr, uint8_t atten_id);       /* ref :170 */
int  perseus_set_attenuator_in_db(perseus_descr *descr, int att_level_in_db);          /* ref :180 */
int  perseus_get_attenuator_values(perseus_descr *descr, int *buf, unsigned int size); /* ref :192 */
int  perseus_set_attenuator_n(perseus_descr *descr, int nlo);              /* ref :203 */
int  perseus_set_adc(perseus_descr *descr, int enableDither, int enablePreamp);        /* ref :218 */
int  perseus_set_ddc_center_freq(perseus_descr *descr, double center_freq_hz,
                                 int enablePresel);                        /* ref :232 */
int  perseus_start_async_input(perseus_descr *descr, uint32_t buffersize,
                               perseus_input_callback callback, void *cb_extra);       /* ref :247 */
int  perseus_stop_async_input(perseus_descr *descr);                       /* ref :257 */
int  perseus_set_sampling_rate(perseus_descr *descr, int sample_rate_value);           /* ref :267 */
int  perseus_set_sampling_rate_n(perseus_descr *descr, unsigned int sample_rate_ordinal); /* ref :277 */
int  perseus_get_sampling_rates(perseus_descr *descr, int *buf, unsigned int size);    /* ref :294 */
int  perseus_is_preserie(perseus_descr *descr, int *flag);                 /* ref :312 */
char *perseus_errorstr(void);                                              /* ref :366 */

/* ---- error codes (reference perseus-sdr.h:317-343) --------------------------- */
#define PERSEUS_NOERROR          0
#define PERSEUS_INVALIDDEV      -1
#define PERSEUS_NULLDESCR       -2
#define PERSEUS_ALREADYOPEN     -3
#define PERSEUS_LIBUSBERR       -4
#define PERSEUS_DEVNOTOPEN      -5
#define PERSEUS_DEVCONF         -6
#define PERSEUS_DEVCLAIMINT     -7
#define PERSEUS_DEVALTINT       -8
#define PERSEUS_FNNOTAVAIL      -9
#define PERSEUS_DEVNOTFOUND     -10
#define PERSEUS_EEPROMREAD      -11
#define PERSEUS_FILENOTFOUND    -12
#define PERSEUS_IOERROR         -13
#define PERSEUS_INVALIDHEXREC   -14
#define PERSEUS_INVALIDEXTREC   -15
#define PERSEUS_FWNOTLOADED     -16
#define PERSEUS_FPGACFGERROR    -17
#define PERSEUS_FPGANOTCFGD     -18
#define PERSEUS_ASYNCSTARTED    -19
#define PERSEUS_NOMEM           -20
#define PERSEUS_CANTCREAT       -21
#define PERSEUS_ERRPARAM        -22
#define PERSEUS_MUTEXIN         -23
#define PERSEUS_BUFFERSIZE      -24
#define PERSEUS_ATTERROR        -25
#define PERSEUS_SNNOTAVAILABLE  -26

/* ---- globals and logging macros (reference perseus-sdr.h:345-364) ------------
 * Clients of the reference use these macros directly (perseustest.c:116), so the
 * three symbols are exported and the macros keep their names and behaviour:
 * "perseus: " / "perseus error: " prefixes on stderr, newline appended.          */
extern int  perseus_dbg_level;
extern char perseus_error_str[1024];
extern int  perseus_error;

#define dbgprintf(level, ...)                              \
    do {                                                   \
        if (perseus_dbg_level >= (level)) {                \
            fputs("perseus: ", stderr);                    \
            fprintf(stderr, __VA_ARGS__);                  \
            fputc('\n', stderr);                           \
        }                                                  \
    } while (0)

#define errorset(code, ...)                                                        \
    (snprintf(perseus_error_str, sizeof(perseus_error_str) - 1, __VA_ARGS__),      \
     (perseus_dbg_level >= 1 ? fprintf(stderr, "perseus error: %s\n", perseus_error_str) : 0), \
     (perseus_error = (code)))

#define errornone(value) (perseus_error = 0, (value))

#ifdef __cplusplus
}
#endif
#endif /* PERSEUS_SDR_AMD_H */
