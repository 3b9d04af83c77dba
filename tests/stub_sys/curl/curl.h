/* syntax-check stand-in for libcurl's public header, the handful of names rtlsdr_ft8d.c uses (see ../README.md) */
#ifndef STUB_CURL_H
#define STUB_CURL_H
typedef void CURL;
typedef enum { CURLE_OK = 0 } CURLcode;
typedef enum { CURL_FORMADD_OK = 0 } CURLFORMcode;
typedef enum { CURLOPT_URL = 10002, CURLOPT_HTTPPOST = 10024 } CURLoption;
typedef enum { CURLFORM_COPYNAME = 1, CURLFORM_COPYCONTENTS = 4, CURLFORM_END = 17 } CURLformoption;
struct curl_httppost;
#define CURL_GLOBAL_ALL 3L
CURLcode curl_global_init(long flags);
CURL *curl_easy_init(void);
CURLcode curl_easy_setopt(CURL *handle, CURLoption option, ...);
CURLcode curl_easy_perform(CURL *handle);
void curl_easy_cleanup(CURL *handle);
const char *curl_easy_strerror(CURLcode code);
CURLFORMcode curl_formadd(struct curl_httppost **first, struct curl_httppost **last, ...);
void curl_formfree(struct curl_httppost *form);
#endif
