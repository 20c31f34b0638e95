/* sanitizer build only: entry points of the device library the stand-in does not implement (never reached by the extract tests) */
#include <stdint.h>
#include <stddef.h>
int md_comm_close(void *c) { (void)c; return -1; }
int md_comm_open_rank() { return -1; }
int md_comm_result_header() { return -1; }
int md_comm_result_recv() { return -1; }
int md_comm_result_send() { return -1; }
int md_comm_wait() { return -1; }
int md_dev_mbias_read() { return -1; }
int md_dev_mbias_submit() { return -1; }
int md_dev_mbias_submit_raw() { return -1; }
int md_dev_perread_download() { return -1; }
int md_dev_perread_download_raw() { return -1; }
int md_dev_perread_submit() { return -1; }
int md_dev_perread_submit_raw() { return -1; }
